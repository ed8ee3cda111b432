"""Hash of the kernel text the bench's PMC traffic numbers belong to (the headline kernel and what it includes).
bench.py quotes profiles/pmc_traffic.json only when its `kernel_source_sha` equals this; tools/prof_bench.sh <tag> pmc rewrites it."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["pow2_kernel.h", "butterflies.h", "device_common.h", "kernels_pow2.hip"]


def sha():
    h = hashlib.sha256()
    for f in FILES:
        h.update(open(os.path.join(ROOT, "ndrustfft_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(sha())
