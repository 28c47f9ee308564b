"""Host boundary of the frame loop (video_unscreen_amd/hostio.py): every way of ordering the copies gives the bytes of the device-resident
loop - the default "hostwait" (a helper thread waits for the compute event, then enqueues the download) and the event-ordered "own"."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clip(H, W, K):
    from video_unscreen_amd.synth import SyntheticClip
    clip = SyntheticClip(H, W, K, config_id=2, device="cuda")
    clip.gen.manual_seed(77)
    return clip, [clip.frame(t) for t in range(K)]


# (135, 241): odd sizes, the parts of a group sit on padded 16-byte offsets; with_seg: the masks come from the human-parsing network,
# 4 frames per forward, into the pipeline's three sets of mask buffers (21 frames: a ragged last group)
@pytest.mark.parametrize("H,W,with_seg", [(270, 480, False), (135, 241, False), (270, 480, True)])
def test_host_loop_equals_device_loop(H, W, with_seg):
    """Inputs staged by copy streams must not let the front / segmentation streams run ahead of the back half into buffers it still
    reads (pipeline slots, mask sets): the staged loop gives the bytes of the loop over device-resident inputs, in every mode."""
    from video_unscreen_amd import hostio
    from video_unscreen_amd.pipeline import BgStepPipeline, default_config
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    K = 21                                                      # more frames than the output ring (8) and the staging ring
    seg_agent = None
    if with_seg:
        from video_unscreen_amd.binseg import HumanSegAgent
        from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
        seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
    pipe = BgStepPipeline(default_config(), H, W, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=4 if with_seg else 1)
    clip, fr = _clip(H, W, K)
    want = []
    pipe.run(iter([(f["frame"], None if with_seg else f["seg"], f["bg"], clip.bg_always) for f in fr]),
             on_result=lambda n, out: want.append((out["alpha"].clone(), out["fg"].clone())))
    torch.cuda.synchronize()
    assert len(want) == K
    packed = [hostio.pack_host_group(f["frame"].cpu(), f["bg"].cpu(), None if with_seg else f["seg"].cpu()) for f in fr]
    loose = [(f["frame"].cpu().pin_memory(), None if with_seg else f["seg"].cpu().pin_memory(), f["bg"].cpu().pin_memory()) for f in fr]
    depth = 14 if with_seg else 9
    for mode in ("hostwait", "own"):
        for groups, outs in ((packed, hostio.packed_outputs(K, H, W)),
                             (loose, (torch.zeros((K, H, W), dtype=torch.uint8).pin_memory(), torch.zeros((K, H, W, 3), dtype=torch.uint8).pin_memory()))):
            io = hostio.HostFrameIO(pipe, depth=depth, down_mode=mode)
            outs[0].zero_(); outs[1].zero_()
            pipe.reset()
            n = pipe.run(io.inputs(groups, clip.bg_always), on_result=io.collect(*outs))
            io.drain()
            io.close()
            assert n == K
            for t in range(K):
                assert torch.equal(outs[0][t], want[t][0].cpu()), (mode, t)
                assert torch.equal(outs[1][t], want[t][1].cpu()), (mode, t)


def test_download_thread_failure_reaches_the_loop():
    """A copy the helper thread cannot enqueue (here: a destination of the wrong size) fails whoever waits on it - the frame loop, when
    it wants the output ring entry back - instead of leaving it waiting."""
    from video_unscreen_amd import hostio
    from video_unscreen_amd.pipeline import BgStepPipeline, default_config
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W, K = 135, 240, 12
    pipe = BgStepPipeline(default_config(), H, W, state_dict=synthetic_state_dict(0))
    clip, fr = _clip(H, W, K)
    groups = [hostio.pack_host_group(f["frame"].cpu(), f["bg"].cpu(), f["seg"].cpu()) for f in fr]
    io = hostio.HostFrameIO(pipe)
    bad_alpha = torch.zeros((K, H, W + 1), dtype=torch.uint8).pin_memory()
    fg = torch.zeros((K, H, W, 3), dtype=torch.uint8).pin_memory()
    with pytest.raises(RuntimeError, match="download thread"):
        pipe.run(io.inputs(groups, clip.bg_always), on_result=io.collect(bad_alpha, fg))
        io.drain()
        for p in (io._last_down,):
            p.synchronize()
    with pytest.raises(RuntimeError, match="download thread"):
        io.close()                                   # the thread's first error is kept and raised again here
    io.close()                                       # ... once
    torch.cuda.synchronize()
    # a clip shorter than the output ring: no later frame ever waits on the failed copies - drain() must still report them
    # (round 3 returned the frame count with stale host outputs and no error: ADVICE round 3)
    pipe.reset()
    io = hostio.HostFrameIO(pipe)
    with pytest.raises(RuntimeError, match="download thread"):
        n = pipe.run(io.inputs(groups[:3], clip.bg_always), on_result=io.collect(bad_alpha, fg))
        assert n == 3
        io.drain()
    io.close()
    torch.cuda.synchronize()


@pytest.mark.parametrize("H,W,K", [(1080, 1920, 27), (2160, 3840, 15)])       # BASELINE.json configs[1] and configs[3] sizes
def test_full_size_loops_agree_sync_pipelined_host(H, W, K):
    """At the bench's sizes with the human-parsing network in the loop (three streams + two copy streams + the download thread):
    the synchronous loop (step()), the pipelined loop over HBM-resident inputs and the host -> host loop give the same bytes."""
    from video_unscreen_amd import hostio
    from video_unscreen_amd.binseg import HumanSegAgent
    from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
    from video_unscreen_amd.pipeline import BgStepPipeline, default_config
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
    pipe = BgStepPipeline(default_config(), H, W, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=12)
    clip, fr = _clip(H, W, K)
    want = []
    for f in fr:
        o = pipe.step(f["frame"], None, f["bg"], clip.bg_always)
        want.append((o["alpha"].clone(), o["fg"].clone()))
    torch.cuda.synchronize()
    for rep in range(3):
        got = []
        pipe.reset()
        pipe.run(iter([(f["frame"], None, f["bg"], clip.bg_always) for f in fr]), on_result=lambda n, o: got.append((o["alpha"].clone(), o["fg"].clone())))
        torch.cuda.synchronize()
        bad = [n for n in range(K) if not (torch.equal(got[n][0], want[n][0]) and torch.equal(got[n][1], want[n][1]))]
        assert not bad, ("pipelined loop, pass %d" % rep, bad)
    groups = [hostio.pack_host_group(f["frame"].cpu(), f["bg"].cpu(), None) for f in fr]
    a_h, f_h = hostio.packed_outputs(K, H, W)
    io = hostio.HostFrameIO(pipe)
    for rep in range(3):
        a_h.zero_(); f_h.zero_()
        pipe.reset()
        pipe.run(io.inputs(groups, clip.bg_always), on_result=io.collect(a_h, f_h))
        io.drain()
        bad = [n for n in range(K) if not (torch.equal(a_h[n], want[n][0].cpu()) and torch.equal(f_h[n], want[n][1].cpu()))]
        assert not bad, ("host loop, pass %d" % rep, bad)
    # the same loop with remove_invalid_objects on the host worker threads (bench.py's host_object_filter leg)
    pipe.object_filter, pipe._filter_choice = "host", "host"
    a_h.zero_(); f_h.zero_()
    pipe.reset()
    pipe.run(io.inputs(groups, clip.bg_always), on_result=io.collect(a_h, f_h))
    io.drain()
    bad = [n for n in range(K) if not (torch.equal(a_h[n], want[n][0].cpu()) and torch.equal(f_h[n], want[n][1].cpu()))]
    assert not bad, ("host loop, host object filter", bad)
    io.close()


def test_slow_reader_small_rings_and_a_consumer_that_never_reports():
    """The paced upload path under awkward conditions: a reader that delivers a frame every few milliseconds (the look-ahead runs dry,
    frames go up on demand), the smallest staging ring the loop admits, several runs through one HostFrameIO - same bytes as the
    device-resident loop; and a loop that never hands results back runs out of staging slots with an error, not a hang."""
    import time
    from video_unscreen_amd import hostio
    from video_unscreen_amd.pipeline import BgStepPipeline, default_config
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W, K = 135, 240, 17
    pipe = BgStepPipeline(default_config(), H, W, state_dict=synthetic_state_dict(0))
    clip, fr = _clip(H, W, K)
    want = []
    pipe.run(iter([(f["frame"], f["seg"], f["bg"], clip.bg_always) for f in fr]), on_result=lambda n, o: want.append((o["alpha"].clone(), o["fg"].clone())))
    torch.cuda.synchronize()
    groups = [hostio.pack_host_group(f["frame"].cpu(), f["bg"].cpu(), f["seg"].cpu()) for f in fr]

    def slow(gs):
        for k, g in enumerate(gs):
            if k % 3 == 0:
                time.sleep(0.004)
            yield g

    a_h, f_h = hostio.packed_outputs(K, H, W)
    io = hostio.HostFrameIO(pipe, depth=6)                       # look-ahead 1, six staging slots
    assert io.lookahead == 1
    for rep in range(3):
        a_h.zero_(); f_h.zero_()
        pipe.reset()
        assert pipe.run(io.inputs(slow(groups), clip.bg_always), on_result=io.collect(a_h, f_h)) == K
        io.drain()
        for t in range(K):
            assert torch.equal(a_h[t], want[t][0].cpu()) and torch.equal(f_h[t], want[t][1].cpu()), (rep, t)
    io.close()
    io2 = hostio.HostFrameIO(pipe, depth=6)
    pipe.reset()
    with pytest.raises(RuntimeError, match="depth 6 is smaller"):
        pipe.run(io2.inputs(groups, clip.bg_always))            # nobody marks a staging slot free
    torch.cuda.synchronize()


@pytest.mark.parametrize("with_seg,K", [(False, 7), (True, 9)])
def test_interleaved_ranges_equal_the_ranges_run_on_their_own(with_seg, K):
    """BgStepPipeline(interleave=2).run_interleaved: two independent frame ranges (two sub-ranges of a shard, each its own alpha_pre chain
    like two of the reference's --range runs: tools/unscreen/bg_offline.py:34-35, 184-192) share every matting forward - and every frame
    of either range has the bits run() gives that range on its own (the batch-2 network is planned as for one image)."""
    from video_unscreen_amd.pipeline import BgStepPipeline, default_config
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W = 270, 480
    seg_agent = None
    if with_seg:
        from video_unscreen_amd.binseg import HumanSegAgent
        from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
        seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
    clip, fr = _clip(H, W, 2 * K)
    items = [(f["frame"], None if with_seg else f["seg"], f["bg"], clip.bg_always) for f in fr]
    ranges = (items[:K], items[K:])
    single = BgStepPipeline(default_config(), H, W, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=4 if with_seg else 1)
    want = []
    for rg in ranges:
        single.reset()
        got = []
        single.run(iter(rg), on_result=lambda n, o: got.append((o["alpha"].clone(), o["fg"].clone())))
        torch.cuda.synchronize()
        want.append(got)
    pair = BgStepPipeline(default_config(), H, W, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=4 if with_seg else 1, interleave=2)
    res = {}
    n = pair.run_interleaved(iter(ranges[0]), iter(ranges[1]), on_result=lambda lane, i, o: res.__setitem__((lane, i), (o["alpha"].clone(), o["fg"].clone())))
    torch.cuda.synchronize()
    assert n == 2 * K and len(res) == 2 * K
    for lane in (0, 1):
        for i in range(K):
            assert torch.equal(res[(lane, i)][0], want[lane][i][0]) and torch.equal(res[(lane, i)][1], want[lane][i][1]), (lane, i)
    # the plain loop of the same pipeline object is untouched by the second network
    pair.reset()
    again = []
    pair.run(iter(ranges[0]), on_result=lambda n, o: again.append(o["alpha"].clone()))
    torch.cuda.synchronize()
    assert all(torch.equal(a, w[0]) for a, w in zip(again, want[0]))
