"""ISA-level checks of the shipped conv kernels, on the CPU: the gfx950 code objects embedded in libunscreen_hip.so are extracted and
disassembled (llvm-objdump; tools/dev/isa_check.py) and two properties are asserted that no run-time test pins deterministically:

1. The staged-image hand-over of conv_wino_kernel: a wave can only wait for its OWN global -> LDS requests, and its
   partner reads the image behind a bare s_barrier, so every `buffer_load ... lds` must be covered by an `s_waitcnt vmcnt(N)` of the
   issuing wave (N <= the vector-memory instructions issued after it: the counter is in order) before the wave has passed two
   barriers (four in the transposed form, which requests two steps ahead).  The race this rule describes lived in rounds 1 - 2 and
   showed once in ~20 000 launches beside another stream; profiles/r04/isa_selftest.txt shows the rule flagging a build with the
   wait removed (tools/dev/isa_selftest.sh).
2. No scratch (spill) instruction inside a loop that contains v_mfma: one spilled value reloaded in a K loop costs ~2000 cycles per
   step, because its wait (vmcnt, in order) is a wait for every load in flight.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
SO = os.path.join(ROOT, "video_unscreen_amd", "libunscreen_hip.so")


@pytest.fixture(scope="module")
def kernels():
    import isa_check as ic
    if not os.path.exists(SO):
        import __graft_entry__ as ge
        ge.build_native()
    out = {}
    for img in ic.code_objects(SO):
        out.update(ic.disassemble(img))
    assert out, "no gfx950 code object found in libunscreen_hip.so"
    return out


def test_every_conv_kernel_is_in_the_library(kernels):
    names = [n for n in kernels if "conv_wino_kernel" in n or "conv_mfma_kernel" in n]
    wino = [n for n in names if "conv_wino_kernel" in n]
    assert len(wino) == 15 and len(names) - len(wino) >= 20, sorted(names)       # (KW 1 | 2 | 4) x IN_RELU x (rect, linear) + transposed x IN_RELU + one run-mode variant
    import isa_check as ic
    for n in names:
        f = ic.facts(kernels[n])
        assert f["mfma"] >= 16 and f["mfma_loops"], n                            # f32 MFMA kernels with a K loop (16 NT MFMAs per step in the direct kernel)
    for n in wino:
        assert ic.facts(kernels[n])["lds_dma"] >= 12, n                          # global -> LDS staging without registers


def test_no_scratch_inside_loops_that_hold_mfma(kernels):
    import isa_check as ic
    bad = {n: ic.facts(ins)["scratch_in_mfma_loops"] for n, ins in kernels.items() if "conv_" in n}
    bad = {n: c for n, c in bad.items() if c}
    assert not bad, bad


def test_winograd_staged_image_hand_over_is_covered_by_a_wait(kernels):
    import isa_check as ic
    checked = 0
    for n, ins in kernels.items():
        if "conv_wino_kernel" not in n:
            continue
        t3 = re.search(r"<\d+, \w+, 2(, \w+)?>", n) is not None
        bad = ic.handover_violations(ins, 4 if t3 else 2)
        assert not bad, (n, [(k, ins[k][1], why) for k, why in bad[:4]])
        checked += 1
    assert checked == 15


def test_direct_kernel_staged_chunk_is_covered_before_its_fragment_reads(kernels):
    """conv_mfma_kernel (round 5): the A operand goes global -> LDS without registers, one K step ahead, into the wave's OWN image, and
    the fragment reads are inline assembly the compiler does not order against the requests: every `buffer_load ... lds` must be covered by
    an `s_waitcnt vmcnt(N)` (N <= the vector-memory instructions issued after it) before the next `ds_read_b128` of the issuing wave."""
    import isa_check as ic
    checked = 0
    for n, ins in kernels.items():
        if "conv_mfma_kernel" not in n:
            continue
        f = ic.facts(ins)
        assert f["lds_dma"] >= 8, (n, f["lds_dma"])                 # four requests in the prologue, four per K step
        bad = ic.handover_violations(ins, 1, stop_at=("ds_read_b128",))
        assert not bad, (n, [(k, ins[k][1], why) for k, why in bad[:4]])
        checked += 1
    assert checked >= 20


def test_hand_over_rule_flags_a_missing_wait():
    """The rule itself on a hand-written instruction stream: request, barrier, barrier without a covering wait -> flagged; with
    `s_waitcnt vmcnt(0)` before the second barrier, or vmcnt(4) behind four newer loads -> clean."""
    import isa_check as ic

    def stream(mid):
        body = [("buffer_load_dwordx4", "v1, s[4:7], s2 offen lds")] + mid + [("s_barrier", ""), ("ds_read_b128", "v[2:5], v6"), ("s_endpgm", "")]
        return [(4 * k, op, a) for k, (op, a) in enumerate(body)]

    assert ic.handover_violations(stream([("s_barrier", ""), ("ds_read_b128", "v[2:5], v6")]), 2)
    assert ic.handover_violations(stream([("ds_read_b128", "v[2:5], v6"), ("s_waitcnt", "vmcnt(0)")]), 2, stop_at=("ds_read_b128",))      # read before the wait
    assert not ic.handover_violations(stream([("s_waitcnt", "vmcnt(0)"), ("ds_read_b128", "v[2:5], v6")]), 2, stop_at=("ds_read_b128",))
    assert not ic.handover_violations(stream([("s_barrier", ""), ("s_waitcnt", "vmcnt(0)")]), 2)
    four = [("buffer_load_dwordx4", "v[8:11], v7, s[8:11], 0 offen")] * 4
    assert not ic.handover_violations(stream(four + [("s_waitcnt", "vmcnt(4)")]), 2)
    assert ic.handover_violations(stream(four[:3] + [("s_waitcnt", "vmcnt(4)"), ("s_barrier", "")]), 2)   # three newer loads: vmcnt(4) does not cover the request
