// aec_cli.cpp -- `aec`: file front-end over the libaec ABI of this library (libaec.so.0).
//
// Same command line as the reference's tool (reference src/aec.c:72-239: options -3 -N -b -d -j -m -n
// -p -r -s -t, SOURCE DEST; defaults 8 bits, block 8, rsi 2, preprocessor on), so that scripts written
// for it run unchanged -- BASELINE config 1 is `aec -d -n16 -j64 -r256 -m typical.rz out`.  The work
// is done by aec_encode / aec_decode, i.e. on the GPU; this file only moves bytes between files and the
// stream object: input is read in pieces of the buffer size, every piece is offered until it is
// consumed, produced bytes are written as they come, and the coder is called with an empty input
// until it delivers nothing more (encode: one AEC_FLUSH call finishes the stream).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/libaec.h"

namespace {

struct Options {
    aec_stream strm{};
    bool decode = false;
    size_t buffer = (size_t)10 << 20;          // bytes per piece
    std::string source, dest;
};

int usage(const char *prog)
{
    fprintf(stderr,
            "usage: %s [OPTION]... SOURCE DEST\n"
            "  encode SOURCE to DEST with CCSDS 121.0-B-2 adaptive entropy coding (MI355X), or decode with -d\n"
            "  -3        24 bit samples are stored in 3 bytes\n"
            "  -N        disable pre/post processing\n"
            "  -b size   internal buffer size in bytes\n"
            "  -d        decode SOURCE (default: encode)\n"
            "  -j n      block size in samples\n"
            "  -m        samples are MSB first (default: LSB)\n"
            "  -n bits   bits per sample\n"
            "  -p        pad RSI to byte boundary\n"
            "  -r n      reference sample interval in blocks\n"
            "  -s        samples are signed (default: unsigned)\n"
            "  -t        use restricted set of code options\n",
            prog);
    return 1;
}

// "-n16" and "-n 16" alike
bool number_arg(int argc, char **argv, int &i, unsigned long &value)
{
    const char *text = argv[i] + 2;
    if (*text == '\0') {
        if (i + 1 >= argc) return false;
        text = argv[++i];
    }
    char *end = nullptr;
    value = strtoul(text, &end, 10);
    return end != text && *end == '\0';
}

bool parse(int argc, char **argv, Options &o)
{
    o.strm.bits_per_sample = 8;
    o.strm.block_size = 8;
    o.strm.rsi = 2;
    o.strm.flags = AEC_DATA_PREPROCESS;
    std::vector<std::string> files;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || a[1] == '\0') {
            files.push_back(a);
            continue;
        }
        unsigned long v = 0;
        switch (a[1]) {
        case '3': o.strm.flags |= AEC_DATA_3BYTE; break;
        case 'N': o.strm.flags &= ~(unsigned)AEC_DATA_PREPROCESS; break;
        case 'd': o.decode = true; break;
        case 'm': o.strm.flags |= AEC_DATA_MSB; break;
        case 'p': o.strm.flags |= AEC_PAD_RSI; break;
        case 's': o.strm.flags |= AEC_DATA_SIGNED; break;
        case 't': o.strm.flags |= AEC_RESTRICTED; break;
        case 'b': if (!number_arg(argc, argv, i, v) || v == 0) return false; o.buffer = v; break;
        case 'j': if (!number_arg(argc, argv, i, v)) return false; o.strm.block_size = (unsigned)v; break;
        case 'n': if (!number_arg(argc, argv, i, v)) return false; o.strm.bits_per_sample = (unsigned)v; break;
        case 'r': if (!number_arg(argc, argv, i, v)) return false; o.strm.rsi = (unsigned)v; break;
        default: return false;
        }
    }
    if (files.size() != 2) return false;
    o.source = files[0];
    o.dest = files[1];
    return true;
}

struct File {
    FILE *f = nullptr;
    ~File() { if (f) fclose(f); }
};

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    if (!parse(argc, argv, o)) return usage(argv[0]);
    File in, out;
    if (!(in.f = fopen(o.source.c_str(), "rb"))) {
        fprintf(stderr, "%s: cannot open %s for reading\n", argv[0], o.source.c_str());
        return 1;
    }
    if (!(out.f = fopen(o.dest.c_str(), "wb"))) {
        fprintf(stderr, "%s: cannot open %s for writing\n", argv[0], o.dest.c_str());
        return 1;
    }
    // whole samples per piece, so that an encoder piece never ends inside a sample
    size_t piece = o.buffer;
    const unsigned bps = o.strm.bits_per_sample;
    const size_t sample = bps > 16 ? ((bps <= 24 && (o.strm.flags & AEC_DATA_3BYTE)) ? 3 : 4) : (bps > 8 ? 2 : 1);
    if (!o.decode) {
        piece = piece / sample * sample;
        if (piece == 0) piece = sample;
    }
    // (the decoder reports AEC_MEM_ERROR when it is left with room for a fraction of a sample, reference
    // decode.c:821-823: the output buffer holds whole samples)
    size_t room = o.buffer > 4096 ? o.buffer : 4096;
    room -= room % sample;
    std::vector<unsigned char> ibuf(piece), obuf(room);

    int rc = o.decode ? aec_decode_init(&o.strm) : aec_encode_init(&o.strm);
    if (rc != AEC_OK) {
        fprintf(stderr, "%s: initialisation failed (%d)%s\n", argv[0], rc,
                rc == AEC_MEM_ERROR ? " -- no usable HIP device?" : "");
        return 1;
    }
    auto step = [&](int flush) -> int {                 // one call; writes what it produced; bytes produced or -1
        o.strm.next_out = obuf.data();
        o.strm.avail_out = obuf.size();
        rc = o.decode ? aec_decode(&o.strm, flush) : aec_encode(&o.strm, flush);
        if (rc != AEC_OK) return -1;
        const size_t n = obuf.size() - o.strm.avail_out;
        if (n && fwrite(obuf.data(), 1, n, out.f) != n) {
            rc = AEC_STREAM_ERROR;
            return -1;
        }
        return (int)(n != 0);
    };
    bool eof = false;
    while (!eof && rc == AEC_OK) {
        const size_t got = fread(ibuf.data(), 1, ibuf.size(), in.f);
        eof = got < ibuf.size();
        o.strm.next_in = ibuf.data();
        o.strm.avail_in = got;
        // offer the piece until it is consumed (output space is renewed on every call)
        size_t before;
        do {
            before = o.strm.avail_in;
            const int progressed = step(AEC_NO_FLUSH);
            if (progressed < 0) break;
            if (!progressed && o.strm.avail_in == before) break;     // needs more input than is here
        } while (o.strm.avail_in);
        if (rc == AEC_OK && o.strm.avail_in && !o.decode && eof && o.strm.avail_in < sample) o.strm.avail_in = 0;   // odd tail bytes
    }
    if (rc == AEC_OK) {
        // drain: empty input until nothing more comes out; the encoder is told that this is the end
        o.strm.avail_in = 0;
        int progressed;
        do {
            progressed = step(o.decode ? AEC_NO_FLUSH : AEC_FLUSH);
        } while (progressed > 0);
    }
    const int end_rc = o.decode ? aec_decode_end(&o.strm) : aec_encode_end(&o.strm);
    if (rc == AEC_OK) rc = end_rc;
    if (rc != AEC_OK) {
        fprintf(stderr, "%s: %s failed (%d)\n", argv[0], o.decode ? "decoding" : "encoding", rc);
        return 1;
    }
    return 0;
}
