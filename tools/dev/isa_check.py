"""ISA-level facts about the shipped conv kernels, read from the gfx950 code objects embedded in libunscreen_hip.so (no GPU needed):
per kernel the loops (backward branches), which of them contain v_mfma, scratch instructions inside them, and where the
s_waitcnt vmcnt(N) / s_barrier pairs sit.  Used by tests/test_isa_cpu.py; run as a script it prints the table.

  python tools/dev/isa_check.py [path to libunscreen_hip.so]
"""
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path, arch="gfx950"):
    """The device ELF images of `arch` inside the library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out_bin = os.path.join(td, "fatbin.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so_path, out_bin], check=True)
        raw = open(out_bin, "rb").read()
    out, pos = [], 0
    while True:
        i = raw.find(MAGIC, pos)
        if i < 0:
            return out
        num = struct.unpack_from("<Q", raw, i + 24)[0]
        off = i + 32
        for _ in range(num):
            o, s, ts = struct.unpack_from("<QQQ", raw, off)
            off += 24
            triple = raw[off:off + ts].decode()
            off += ts
            if arch in triple and s > 0:
                out.append(raw[i + o:i + o + s])
        pos = i + 24


def disassemble(image):
    """-> {demangled kernel name: [(address, mnemonic, operands)]}"""
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".o") as f:
        f.write(image)
        f.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], check=True, capture_output=True, text=True).stdout
    names = sorted(set(re.findall(r"^[0-9a-f]+ <([^>]+)>:$", txt, re.M)))
    dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines())) if names else {}
    kernels, cur = {}, None
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", ln)
        if m:
            cur = kernels.setdefault(dem.get(m.group(1), m.group(1)), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):[^<]*(<.*>)?", ln)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2) + (" " + m.group(4) if m.group(4) else "")))
    return kernels


def loops(ins):
    """Backward branches -> [(index of the target, index of the branch)]; the branch target is the operand's absolute address when
    llvm-objdump prints one (`s_cbranch_scc1 65412 <sym+0x1a4>` or a plain number = signed dword offset from the next instruction)."""
    addr_idx = {a: k for k, (a, _, _) in enumerate(ins)}
    out = []
    for k, (a, op, args) in enumerate(ins):
        if not op.startswith(("s_cbranch", "s_branch")):
            continue
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", args)
        tgt = None
        if m and ins:
            base = ins[0][0]
            tgt = base + int(m.group(1), 16)
        else:
            m = re.match(r"(-?\d+)", args)
            if m:
                off = int(m.group(1))
                if off >= 32768:
                    off -= 65536
                tgt = a + 4 + 4 * off
        if tgt is not None and tgt <= a and tgt in addr_idx:
            out.append((addr_idx[tgt], k))
    return out


def facts(ins):
    lp = loops(ins)
    # innermost loops that hold MFMAs (a K loop; not the item loop of a run-mode kernel, which holds the whole body)
    mf = [(t, b) for t, b in lp if any(op.startswith("v_mfma") for _, op, _ in ins[t:b + 1])
          and not any((t2, b2) != (t, b) and t <= t2 and b2 <= b and (t2 > t or b2 < b) for t2, b2 in lp)]
    scratch_in = sum(1 for t, b in mf for _, op, _ in ins[t:b + 1] if op.startswith("scratch_"))
    return {"loops": len(lp), "mfma_loops": mf, "scratch_in_mfma_loops": scratch_in,
            "scratch_total": sum(1 for _, op, _ in ins if op.startswith("scratch_")),
            "mfma": sum(1 for _, op, _ in ins if op.startswith("v_mfma")),
            "lds_dma": sum(1 for _, op, a in ins if op.startswith("buffer_load") and a.rstrip().endswith("lds"))}


if __name__ == "__main__" and "--handover" not in sys.argv:
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "video_unscreen_amd", "libunscreen_hip.so")
    for img in code_objects(so):
        for name, ins in disassemble(img).items():
            if "conv_" not in name:
                continue
            f = facts(ins)
            print(f"{name[:90]:90s} instr {len(ins):6d} mfma {f['mfma']:4d} lds-dma {f['lds_dma']:3d} loops {f['loops']:2d} (with mfma {len(f['mfma_loops'])}) "
                  f"scratch {f['scratch_total']:3d} (inside mfma loops {f['scratch_in_mfma_loops']})")


# ---- the staged-image hand-over of conv_wino_kernel ("who waits for the staged image")
VMEM = ("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic", "scratch_load", "scratch_store", "flat_load", "flat_store")


def _vmcnt(args):
    m = re.search(r"vmcnt\((\d+)\)", args)
    return int(m.group(1)) if m else None


def handover_violations(ins, max_barriers, stop_at=()):
    """A wave can only wait for its OWN global -> LDS requests (the load counter is per wave), and the partner wave reads the staged
    image after a workgroup barrier without any wait of its own.  So every `buffer_load ... lds` must be followed - on the path the
    issuing wave runs - by an `s_waitcnt vmcnt(N)` that covers it (N <= the number of vector-memory instructions issued after the
    request: the counter is in order) BEFORE the wave passes `max_barriers` s_barriers.  The walk follows the layout: straight on,
    an unconditional forward branch is taken, a backward branch whose target lies at or before the request closes the loop once
    (the walk continues at the target), other conditional branches fall through.  stop_at: mnemonic prefixes that must not be reached
    before the covering wait either (the direct conv kernel reads its OWN image: `ds_read_b128`).  -> [(index of the request, what happened)]"""
    lp = {b: t for t, b in loops(ins)}
    addr_idx = {a: k for k, (a, _, _) in enumerate(ins)}
    bad = []
    for k0, (_, op, args) in enumerate(ins):
        if not (op.startswith("buffer_load") and args.rstrip().split("//")[0].rstrip().endswith("lds")):
            continue
        newer, barriers, k, wrapped, steps, ok = 0, 0, k0 + 1, False, 0, None
        while k < len(ins) and steps < 4 * len(ins):
            steps += 1
            _, o, a = ins[k]
            if o == "s_waitcnt":
                n = _vmcnt(a)
                if n is not None and n <= newer:
                    ok = True
                    break
            elif o == "s_barrier":
                barriers += 1
                if barriers >= max_barriers:
                    ok = False
                    break
            elif stop_at and o.startswith(stop_at):
                ok = False
                break
            elif o.startswith(VMEM):
                newer += 1
            elif o == "s_endpgm":
                ok = True          # the wave ends: nobody reads after it without a barrier it would have had to pass
                break
            elif o == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", a)
                tgt = addr_idx.get(ins[0][0] + int(m.group(1), 16)) if m else None
                if tgt is not None and tgt > k:
                    k = tgt
                    continue
            if k in lp and lp[k] <= k0 and not wrapped:
                wrapped = True
                k = lp[k]
                continue
            k += 1
        if not ok:
            bad.append((k0, f"passed {barriers} barrier(s) with {newer} newer vector-memory instruction(s) and no covering vmcnt wait"))
    return bad


if __name__ == "__main__" and "--handover" in sys.argv:
    so = os.path.join(ROOT, "video_unscreen_amd", "libunscreen_hip.so")
    for img in code_objects(so):
        for name, ins in disassemble(img).items():
            if "conv_wino_kernel" in name:
                t3 = re.search(r"<\d+, \w+, 2(, \w+)?>", name) is not None
                v = handover_violations(ins, 4 if t3 else 2)
                print(name[-60:], "lds requests", facts(ins)["lds_dma"], "violations", v[:6])
