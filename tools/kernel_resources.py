#!/usr/bin/env python3
"""Register / scratch budget of every kernel in the built library, read from the gfx950 code objects' metadata notes
(llvm-objdump --offloading + llvm-readelf --notes; no GPU needed).

    python tools/kernel_resources.py [--spills] [path/to/liblr2ppo_hip.so]

-> one line per kernel: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes, LDS bytes.  tests/test_host_cpu.py asserts
that nothing outside ALLOWED_SCRATCH spills."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LR2_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "lr2ppo_amd", "csrc", "liblr2ppo_hip.so")

# kernels that are allowed private memory, and why
ALLOWED_SCRATCH = {
    "ndcg_kernel": "three 64-entry per-thread sort arrays (one thread ranks one ragged item; latency-bound, off the training path)",
}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except (OSError, subprocess.CalledProcessError):
        return names


def kernels(lib_path=LIB):
    """-> list of dicts {name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds} for every kernel of every gfx950 code
    object bundled in the shared library."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], capture_output=True, text=True, check=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True,
                                   text=True, check=True).stdout
            for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                get = lambda k, d=0: int(m.group(1)) if (m := re.search(r"\.%s:\s+(\d+)" % k, blk)) else d     # noqa: E731
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out.append({"mangled": name, "vgpr": get("vgpr_count"), "agpr": get("agpr_count"), "sgpr": get("sgpr_count"),
                            "vgpr_spill": get("vgpr_spill_count"), "sgpr_spill": get("sgpr_spill_count"),
                            "scratch": get("private_segment_fixed_size"), "lds": get("group_segment_fixed_size")})
    for k, n in zip(out, demangle([k["mangled"] for k in out])):
        n = re.sub(r"^void\s+", "", n)
        n = n.replace("(anonymous namespace)::", "").replace("lr2gemm::", "")
        i = n.find("(")
        k["name"] = n[:i] if i > 0 else n
    return out


def main():
    only_spills = "--spills" in sys.argv
    paths = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = kernels(paths[0] if paths else LIB)
    print(f"{len(ks)} kernels")
    for k in sorted(ks, key=lambda k: (-k["scratch"], -k["vgpr"])):
        if only_spills and not (k["vgpr_spill"] or k["sgpr_spill"] or k["scratch"]):
            continue
        print(f"vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d}  spill v {k['vgpr_spill']:4d} s {k['sgpr_spill']:3d}  "
              f"scratch {k['scratch']:5d} B  lds {k['lds']:6d} B  {k['name']}")


if __name__ == "__main__":
    main()
