"""Build libmcq_hip.so in-tree:  python -m neuron_poker_amd.build  (hipcc --offload-arch=gfx950)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False):
    cmd = ["make", "-j4", "-C", os.path.join(HERE, "csrc")]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode:
        sys.stdout.write(res.stdout)
    if res.returncode:
        raise RuntimeError("building libmcq_hip.so failed")
    return os.path.join(HERE, "libmcq_hip.so")


if __name__ == "__main__":
    print(build(verbose=True))
