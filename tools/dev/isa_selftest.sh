#!/bin/bash
# Does the ISA-level hand-over rule (tools/dev/isa_check.py, tests/test_isa_cpu.py) have the power to catch the round-3 bug?  Compile
# usc_conv.hip (device code only, no GPU) twice with one wait removed each time - (a) the vmcnt(0) the h = 1 wave needs before the barrier
# that hands the pair's second image over (the bug that was found in round 3), (b) the explicit vmcnt(4) at the end of the MFMA phase -
# and run the rule on the resulting code objects.  Output: profiles/r04/isa_selftest.txt
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
SRC=$ROOT/video_unscreen_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-slp-vectorize --cuda-device-only -c"
for v in a b; do
  mkdir -p $T/$v/video_unscreen_amd/csrc $T/$v/include
  cp $ROOT/include/unscreen_hip.h $T/$v/include/
  cp $SRC/usc_device.h $SRC/usc_host.h $T/$v/video_unscreen_amd/csrc/
done
python3 - "$SRC/usc_conv.hip" "$T/a/video_unscreen_amd/csrc/usc_conv.hip" "$T/b/video_unscreen_amd/csrc/usc_conv.hip" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
a = re.sub(r"(then multiplied one K step of stale patch values[^\n]*\n)\s*wait_vm0\(\);\n", r"\1", s, count=1)
assert a != s, "pattern (a) not found"
b = s.replace('asm volatile("s_waitcnt vmcnt(4)" ::: "memory");', "", 1)
assert b != s, "pattern (b) not found"
open(sys.argv[2], "w").write(a)
open(sys.argv[3], "w").write(b)
PY
for v in a b; do
  (cd $T/$v/video_unscreen_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -o $T/$v.co usc_conv.hip 2>&1 | grep -i "error" || true)
done
python3 - "$ROOT" "$T" <<'PY'
import re, sys
sys.path.insert(0, sys.argv[1] + "/tools/dev")
import isa_check as ic
for v, what in (("a", "h = 1 wave's vmcnt(0) before the second image's hand-over barrier removed (the round-3 bug)"),
                ("b", "explicit vmcnt(4) at the end of the MFMA phase removed")):
    img = open(f"{sys.argv[2]}/{v}.co", "rb").read()
    if img[:24] == ic.MAGIC:                      # a bundle (host stub + device image): take the device image
        import struct
        num = struct.unpack_from("<Q", img, 24)[0]
        off = 32
        for _ in range(num):
            o, s, ts = struct.unpack_from("<QQQ", img, off)
            off += 24 + ts
            if s > 0:
                dev = img[o:o + s]
        img = dev
    print(f"== {what}")
    for name, ins in ic.disassemble(img).items():
        if "conv_wino_kernel" in name:
            t3 = re.search(r"<\d+, \w+, 2(, \w+)?>", name) is not None
            bad = ic.handover_violations(ins, 4 if t3 else 2)
            print(f"   {name[name.index('conv_wino'):][:45]:45s} {'FLAGGED: %d request(s), first: %s' % (len(bad), bad[0][1]) if bad else 'passes'}")
PY
rm -rf $T
