#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in libgaz_engine.so's gfx950 code objects (llvm-readelf --notes on the
unbundled objects): VGPRs, spilled VGPRs / SGPRs, private (scratch) bytes per lane, LDS.  CPU only.
usage: python tools/kernel_resources.py [filter-substring] [--json out.json]"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
def kernels(obj):
    out = []
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "dev.co"); fb = os.path.join(td, "dev.fatbin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    for blk in txt.split("- .agpr_count:")[1:]:
        def f(k):
            m = re.search(rf"\.{k}:\s+(\S+)", blk); return m.group(1) if m else None
        name = subprocess.check_output(["c++filt", f("name")], text=True).strip()
        out.append(dict(kernel=name, vgpr=int(f("vgpr_count")), agpr=int(blk.split()[0]), sgpr=int(f("sgpr_count")), vgpr_spill=int(f("vgpr_spill_count")),
                        sgpr_spill=int(f("sgpr_spill_count")), scratch_bytes=int(f("private_segment_fixed_size")), lds_static=int(f("group_segment_fixed_size"))))
    return out
if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = args[0] if args and not (len(sys.argv) > 2 and sys.argv[-2] == "--json" and args[0] == sys.argv[-1]) else ""
    rows = []
    for o in ("engine.o", "resnet.o"):
        rows += kernels(os.path.join(ROOT, "grok_alpha_zero_amd", "csrc", "build", o))
    rows = [r for r in rows if flt in r["kernel"]]
    for r in rows:
        print(f'{r["kernel"][:110]:110s} vgpr {r["vgpr"]:3d} agpr {r["agpr"]:3d} sgpr {r["sgpr"]:3d} spill v{r["vgpr_spill"]} s{r["sgpr_spill"]} scratch {r["scratch_bytes"]} B')
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
