import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built libraries are not in the repository's history: a fresh checkout builds them first
    # (hipcc cross-compiles gfx950 without a GPU; the oracle needs gcc only).
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "libaec_amd", "lib", "libaec.so.0")):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(root, "libaec_amd", "csrc")], check=True,
                       stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(root, "oracle", "_build", "libaec_oracle.so")):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, stdout=subprocess.DEVNULL)


GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    """tests/golden/vectors.npz: inputs + bytes produced by the reference (make_golden.py)."""

    def __init__(self):
        z = np.load(os.path.join(GOLDEN_DIR, "vectors.npz"))
        self.names = [str(n) for n in z["names"]]
        self.params = z["params"]
        self.decoded_len = z["decoded_len"]
        self._in, self._out = z["inputs"], z["outputs"]
        self._io, self._oo = z["in_off"], z["out_off"]

    def __len__(self):
        return len(self.names)

    def case(self, i):
        bps, bs, rsi, flags = (int(v) for v in self.params[i])
        return (self.names[i], bps, bs, rsi, flags,
                self._in[int(self._io[i]):int(self._io[i + 1])],
                self._out[int(self._oo[i]):int(self._oo[i + 1])].tobytes(),
                int(self.decoded_len[i]))


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def typical_rz():
    with open(os.path.join(GOLDEN_DIR, "typical.rz"), "rb") as f:
        return f.read()
