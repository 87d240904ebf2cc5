#!/usr/bin/env python3
"""Instructions per phase of a kernel, from an ISA listing with line tables.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -gline-tables-only -S --cuda-device-only \
          -o /tmp/aec_dec_g.s libaec_amd/csrc/aec_dec.hip
    python tests/isa_phases.py /tmp/aec_dec_g.s k_decodeILi16ELi2ELb1ELi2ELb0EE --samples 32

Every instruction of the kernel is attributed to the source line its `.loc` names (the innermost inlined frame) and the
lines to PHASES by the table below (source ranges found by the markers in the sources, so the table follows edits).  The
counts are STATIC: a line that the compiler copied (the loop body holds two blocks of 16 samples; the decoder of a block
exists with and without a reference sample) counts once per copy, paths that a block rarely takes (codes of 31+ zeros,
second attempts, refills, reports) are listed as "rare paths" and left out of the per-sample figures.  --samples is the
number of samples the steady-state loop body covers; the dynamic figure (rocprofv3 --pmc SQ_INSTS_VALU / samples) stands
beside it in profiles/r06/k_decode_isa.txt.
"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libaec_amd", "csrc")


def find(lines, pattern, start=0):
    rx = re.compile(pattern)
    for i in range(start, len(lines)):
        if rx.search(lines[i]):
            return i + 1                      # 1-based
    raise SystemExit(f"marker not found: {pattern}")


def phases_decode(src_dir):
    """(file, first line, last line, phase, hot) for k_decode<16, 2, SEG> -- the path of blocks without a reference sample."""
    lane = open(os.path.join(src_dir, "aec_lane.h")).read().split("\n")
    dec = open(os.path.join(src_dir, "aec_dec.hip")).read().split("\n")
    L = lambda p, s=0: find(lane, p, s)
    D = lambda p, s=0: find(dec, p, s)
    t = []
    noref = L(r"AEC_HD uint32_t decode_block_noref")
    hdr0 = L(r"const uint64_t U0 = peek64\(src, p\);", noref)
    uloop = L(r"for \(uint32_t g0 = 0; g0 < \(uint32_t\)BS; g0 \+= GRP\)", hdr0)
    bad = L(r"if \(AEC_ANY\(bad\)\)", uloop)
    fld = L(r"---- 3\. field phase", bad)
    wide = L(r"if \(AEC_ANY\(kk > 16\)\)", fld)
    med = L(r"else if \(AEC_ANY\(kk > 8\)\)", wide)
    nar = L(r"narrow fields: eight of them per 64-bit peek", med)
    tail = L(r"---- 4\. second extension / zero run", nar)
    end_noref = L(r"^}", tail)
    se_if = L(r"if \(AEC_ANY\(se\)\)", tail)
    zero_if = L(r"if \(AEC_ANY\(zero\)\)", se_if)
    ret = L(r"if \(!live\) return DEC_OK;", zero_if)
    # (the line of a wave-uniform test belongs to the steady-state path, the branch behind it does not)
    t += [("aec_lane.h", noref, uloop - 1, "header: option, selector, k", True),
          ("aec_lane.h", uloop, bad, "unary phase (16 codes from a 64-bit window)", True),
          ("aec_lane.h", bad + 1, fld - 3, "rare: codes of 31+ zeros, one by one", False),
          ("aec_lane.h", fld - 2, fld - 1, "unary phase (16 codes from a 64-bit window)", True),
          ("aec_lane.h", fld, wide, "field phase (k-bit fields, eight per window)", True),
          ("aec_lane.h", wide + 1, med - 1, "rare: fields of more than 8 bits", False),
          ("aec_lane.h", med, med, "field phase (k-bit fields, eight per window)", True),
          ("aec_lane.h", med + 1, nar - 1, "rare: fields of more than 8 bits", False),
          ("aec_lane.h", nar, se_if, "field phase (k-bit fields, eight per window)", True),
          ("aec_lane.h", se_if + 1, zero_if - 1, "second extension / zero blocks (when a lane of the wavefront holds one)", False),
          ("aec_lane.h", zero_if, zero_if, "field phase (k-bit fields, eight per window)", True),
          ("aec_lane.h", zero_if + 1, ret - 1, "second extension / zero blocks (when a lane of the wavefront holds one)", False),
          ("aec_lane.h", ret, end_noref, "header: option, selector, k", True)]
    anyb = L(r"AEC_HD uint32_t decode_block_any")
    t += [("aec_lane.h", anyb, noref - 1, "rare: the block with the reference sample (1 of 64 .. 128)", False)]
    pk = L(r"AEC_HD uint32_t peek32")
    us = L(r"AEC_HD uint32_t unary_slow")
    t += [("aec_lane.h", pk, us - 1, "stream window: 64 bits out of three ring words", True),
          ("aec_lane.h", us, anyb - 1, "rare: codes of 31+ zeros, one by one", False)]
    up = L(r"AEC_HD uint32_t unpp_unsigned")
    se = L(r"AEC_HD bool se_lookup")
    t += [("aec_lane.h", up, se - 1, "inverse predictor, exact steps (blocks near the ends of the range)", None),
          ("aec_lane.h", se, L(r"^}", se), "second extension / zero blocks (when a lane of the wavefront holds one)", False)]
    exact_hot = False
    sb = D(r"__device__ __forceinline__ void store_block\(")
    try:
        fast = D(r"if \(!__any\(!fits\)\)", sb)
        slow = D(r"^        } else {", fast)
    except SystemExit:                          # (before round 6: the exact steps only)
        fast = slow = D(r"uint32_t v\[BS\];", sb)
        exact_hot = True
    pack = D(r"^    if \(BYTES == 4\) {", slow)
    sbe = D(r"generic block size / container: sample by sample", pack)
    t += [("aec_dec.hip", sb, fast - 1, "inverse predictor: sum of the residuals, the test", True),
          ("aec_dec.hip", fast, slow - 1, "inverse predictor: running sum (shift, sign, xor, add)", True),
          ("aec_dec.hip", slow, pack - 1, "inverse predictor, exact steps (blocks near the ends of the range)", None),
          ("aec_dec.hip", pack, sbe - 1, "byte order and packing (v_perm), store to the staging row", True)]
    rs = D(r"struct RingSrc")
    lw = D(r"four consecutive stream words starting at absolute index", rs)
    rp = D(r"__device__ __forceinline__ void ring_put4", lw)
    rpe = D(r"^}", rp)
    w2 = D(r"void word2\(uint32_t i", rs)
    w3 = D(r"// words i, i\+1, i\+2", w2)
    # (two words: the 32-bit peeks of the code-by-code reader and of wide fields)
    t += [("aec_dec.hip", w2, w3 - 1, "rare: codes of 31+ zeros, one by one", False),
          ("aec_dec.hip", rs, lw - 1, "stream window: 64 bits out of three ring words", True),
          ("aec_dec.hip", lw, rp - 1, "ring top-up: the 16-byte loads a block ahead", True),
          ("aec_dec.hip", rp, rpe, "ring top-up: landing the loads (byte swap, LDS writes)", True)]
    kd = D(r"^k_decode\(const Cfg c")
    fl = D(r"auto flush = \[&\]\(uint32_t group\)", kd)
    fle = D(r"^    };", fl)
    loop = D(r"for \(; __any\(b < nb && ok\); b \+= OU \* UNR\)", fle)
    refill = D(r"rare: a lane fell behind", loop)
    blk = D(r"---- one block per lane ----", refill)
    att = D(r"if \(attempt != 0 \|\| needw >= maxw", blk)
    over = D(r"const bool over = parse &&", att)
    land = D(r"if \(uu == 0\) {", over)
    sums = D(r"if \(SUMS\) {", land)
    gen = D(r"} else if \(live\) {", sums)
    kde = D(r"^}", gen)
    t += [("aec_dec.hip", kd, fl - 1, "prologue (once per lane: item, ring fill)", False),
          ("aec_dec.hip", fl, fle, "staging rows written out transposed (LDS read, lane permutes, 64-byte stores)", True),
          ("aec_dec.hip", fle + 1, loop - 1, "prologue (once per lane: item, ring fill)", False),
          ("aec_dec.hip", loop, refill, "loop control, lane state (live, zero runs, produced)", True),
          ("aec_dec.hip", refill + 1, blk - 1, "rare: synchronous refill of a lane that fell behind", False),
          ("aec_dec.hip", blk, att, "loop control, lane state (live, zero runs, produced)", True),
          ("aec_dec.hip", att + 1, over - 1, "rare: second attempt of a long coded data set", False),
          ("aec_dec.hip", over, land - 1, "loop control, lane state (live, zero runs, produced)", True),
          ("aec_dec.hip", land, sums - 1, "ring top-up: landing the loads (byte swap, LDS writes)", True),
          ("aec_dec.hip", sums, gen - 1, "loop control, lane state (live, zero runs, produced)", True),
          ("aec_dec.hip", gen, kde, "rare: generic block sizes", False)]
    # (before round 6 the exact steps were the steady-state path; the signed and the unsigned step are both compiled and
    # one of them runs: the figure counts both)
    t = [(f, lo, hi, ("inverse predictor: exact steps, signed and unsigned variant (one runs)" if exact_hot else ph) if h is None else ph,
          exact_hot if h is None else h) for (f, lo, hi, ph, h) in t]
    # error reports (atomics on the result record, a 64-bit division for the RSI's number): never in a healthy stream
    for i, l in enumerate(dec):
        if re.search(r"\breport\(res|atomicMax\(|atomicOr\(&res|atomicMin\(", l) and kd <= i + 1 <= kde:
            t.insert(0, ("aec_dec.hip", i + 1, i + 1, "rare: error reports", False))
    t.insert(0, ("amd_hip_atomic.h", 0, 10 ** 9, "rare: error reports", False))
    t.insert(0, ("amd_warp_functions.h", 0, 10 ** 9, "staging rows written out transposed (LDS read, lane permutes, 64-byte stores)", True))
    return t


def phases_encode(src_dir):
    """(file, first line, last line, phase, hot) for k_analyze / k_pack: by function (the steady-state path is the packed
    16-bit one of the templated block sizes; the generic loaders and writers are listed as rare)."""
    enc = open(os.path.join(src_dir, "aec_enc.hip")).read().split("\n")
    lane = open(os.path.join(src_dir, "aec_lane.h")).read().split("\n")
    E = lambda p, s=0: find(enc, p, s)
    L = lambda p, s=0: find(lane, p, s)
    t = []

    def fn(lines, fname, start_pat, phase, hot, table):
        a = find(lines, start_pat)
        b = a
        depth = 0
        seen = False
        for i in range(a - 1, len(lines)):                 # to the closing brace of the function
            depth += lines[i].count("{") - lines[i].count("}")
            seen = seen or "{" in lines[i]
            if seen and depth == 0:
                b = i + 1
                break
        table.append((fname, a, b, phase, hot))

    A = "phase A: loads, byte order, mapped residuals (predictor), rows to LDS"
    fn(enc, "aec_enc.hip", r"void wave_lds_fence\(", "LDS fences, wave scans (DPP)", True, t)
    fn(enc, "aec_enc.hip", r"uint32_t wave_scan_dpp\(", "LDS fences, wave scans (DPP)", True, t)
    fn(enc, "aec_enc.hip", r"__forceinline__ Seg seg_geom\(", "segment geometry (64-bit divisions, once per segment)", True, t)
    fn(enc, "aec_enc.hip", r"uint32_t pp_unsigned_pk\(", "rare: the exact mapping (stretches near the ends of the range)", False, t)
    fn(enc, "aec_enc.hip", r"void pp_words_pk\(", A, True, t)
    fn(enc, "aec_enc.hip", r"void fast_issue\(", A, True, t)
    fn(enc, "aec_enc.hip", r"void fast_finish\(", A, True, t)
    fn(enc, "aec_enc.hip", r"void direct_issue\(", A, True, t)
    fn(enc, "aec_enc.hip", r"void direct_finish\(", A, True, t)
    fn(enc, "aec_enc.hip", r"void load_segment_generic\(", "rare: generic loader (ragged ends, odd block sizes)", False, t)
    fn(enc, "aec_enc.hip", r"^struct Feeder \{", A, True, t)
    fn(enc, "aec_enc.hip", r"^struct BlockRegs \{", "block out of its LDS row into registers", True, t)
    fn(enc, "aec_enc.hip", r"bool block_is_zero\(", "option selection (fs(k) by packed shifts and dot products, second extension, choice)", True, t)
    fn(enc, "aec_enc.hip", r"BlockChoice choose_option_pk\(", "option selection (fs(k) by packed shifts and dot products, second extension, choice)", True, t)
    fn(enc, "aec_enc.hip", r"uint32_t analyze_segment\(", "segment: zero runs by ballot, lengths, k clamps, summaries", True, t)
    fn(enc, "aec_enc.hip", r"void emit_segment\(", "segment: prefix sum of lengths, k per block, emission control", True, t)
    for pat, ph, hot in ((r"AEC_HD void assess_split_with\(", "option selection (fs(k) by packed shifts and dot products, second extension, choice)", True),
                         (r"AEC_HD BlockChoice choose_from\(", "option selection (fs(k) by packed shifts and dot products, second extension, choice)", True),
                         (r"AEC_HD uint32_t zero_run_at\(", "segment: zero runs by ballot, lengths, k clamps, summaries", True),
                         (r"^struct BitWriter \{", "bit writer (LDS image of the segment: shifts, atomic or)", True),
                         (r"AEC_HD void emit_block\(", "rare: emission code by code (blocks of more than 16 samples, long codes)", False),
                         (r"AEC_HD bool small_eligible\(", "emission of a block of up to 16 samples: unary and field regions in two 64-bit registers", True),
                         (r"AEC_HD void emit_small\(", "emission of a block of up to 16 samples: unary and field regions in two 64-bit registers", True)):
        fn(lane, "aec_lane.h", pat, ph, hot, t)
    # kernels' own bodies
    fn(enc, "aec_enc.hip", r"^k_analyze\(const Cfg c", "kernel loop: segments of a wave, prefetch of the next, summaries out", True, t)
    fn(enc, "aec_enc.hip", r"^k_pack\(const Cfg c", "kernel loop: segments of a wave, LDS image copied out byte-swapped, shared words", True, t)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("symbol", help="a substring of the kernel's mangled name")
    ap.add_argument("--samples", type=int, default=32, help="samples the steady-state loop body covers")
    ap.add_argument("--lines", action="store_true", help="list the counts per source line as well")
    ap.add_argument("--src", default=CSRC, help="directory of the sources the listing was compiled from")
    ap.add_argument("--encoder", action="store_true", help="the phase table of k_analyze / k_pack instead of k_decode's")
    args = ap.parse_args()
    files = {}
    body = []
    inside = False
    with open(args.asm) as f:
        for l in f:
            m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
            if m:
                files[int(m.group(1))] = os.path.basename(m.group(2))
            if not inside and re.match(r"^[A-Za-z_].*" + re.escape(args.symbol) + r".*:\s*(;.*)?$", l) and not l.startswith("\t"):
                inside = True
            if inside:
                body.append(l.rstrip("\n"))
                if re.match(r"\s*\.size\s", l):
                    break
    if not body:
        raise SystemExit("kernel not found")
    table = phases_encode(args.src) if args.encoder else phases_decode(args.src)
    instr = re.compile(r"^\t([vsd]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+)\s")

    def phase_of(loc):
        for (fn, lo, hi, ph, h) in table:
            if loc[0] == fn and lo <= loc[1] <= hi:
                return ph, h
        return "other (runtime headers, unattributed)", True

    # basic blocks: (depth, [(loc, kind)])
    blocks = []
    cur = ("?", 0)
    depth = 0
    blk = []
    for l in body:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if l.startswith(".LBB") or l.startswith("; %bb"):
            if blk:
                blocks.append((depth, blk))
            blk = []
            d = re.search(r"Depth[= ](\d+)", l)
            depth = int(d.group(1)) if d else 0
            continue
        m = instr.match(l)
        if not m:
            continue
        op = m.group(1)
        kind = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM"
        blk.append((cur, kind))
    if blk:
        blocks.append((depth, blk))
    # A basic block that holds an instruction of a rare path belongs to that path as a whole -- the shared helpers
    # inlined into it (the 64-bit window out of the ring, say) included; so does everything in a loop of depth 3 (the
    # loops of the code-by-code reader) and outside the main loop (prologue, epilogue).
    counts = collections.Counter()
    per_line = collections.Counter()
    for depth, blk in blocks:
        rare = collections.Counter(phase_of(loc)[0] for loc, k in blk if not phase_of(loc)[1])
        for loc, kind in blk:
            phase, hot = phase_of(loc)
            if depth == 0:
                phase, hot = "prologue (once per lane: item, ring fill)", False
            elif rare:
                phase, hot = rare.most_common(1)[0][0], False
            elif depth >= 3:
                phase, hot = "rare: codes of 31+ zeros, one by one", False
            counts[(phase, hot, kind)] += 1
            per_line[(loc, kind, depth)] += 1
    phases = sorted({(p, h) for (p, h, k) in counts}, key=lambda x: (not x[1], x[0]))
    print(f"{'phase':88s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s}   VALU per sample")
    tot = collections.Counter()
    for p, h in phases:
        v, s, l, m = (counts[(p, h, k)] for k in ("VALU", "SALU", "LDS", "VMEM"))
        per = f"{v / args.samples:6.2f}" if h else "     -"
        print(f"{('' if h else '(') + p + ('' if h else ')'):88s} {v:6d} {s:6d} {l:5d} {m:5d}   {per}")
        if h:
            tot.update({"VALU": v, "SALU": s, "LDS": l, "VMEM": m})
    print(f"{'steady-state path, static':88s} {tot['VALU']:6d} {tot['SALU']:6d} {tot['LDS']:5d} {tot['VMEM']:5d}   {tot['VALU'] / args.samples:6.2f}")
    if args.lines:
        print()
        for (c, k, d), n in sorted(per_line.items()):
            if k == "VALU":
                print(f"  {c[0]}:{c[1]:5d} depth {d} {n:5d}")


if __name__ == "__main__":
    sys.exit(main())
