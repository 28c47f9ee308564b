"""Register / scratch / LDS figures of every kernel in libunscreen_hip.so (from the code objects' metadata notes; no GPU needed):
python tools/dev/kernel_regs.py [substring of a kernel name ...]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_check import code_objects, LLVM, ROOT

so = os.environ.get("USC_LIB", os.path.join(ROOT, "video_unscreen_amd", "libunscreen_hip.so"))
pats = sys.argv[1:]
rows = []
for img in code_objects(so):
    with tempfile.NamedTemporaryFile(suffix=".o") as f:
        f.write(img); f.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        rows.append((dem[:110], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count")))
for r in sorted(rows):
    if not pats or any(p in r[0] for p in pats):
        print("%-110s vgpr %4s agpr %4s sgpr %4s scratch %5s lds %6s spills %s" % r)
