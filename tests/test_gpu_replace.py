"""GPU parity tests of the replace loop (reference tools/replace/replace.py:28-82; BASELINE.json configs[4], the compositing half):
the one-launch per-frame kernel against the stand-alone kernels and against the CPU oracle chain, bit for bit, and the tool on
1920 x 1080 raw-array inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _inputs(H, W, sh, sw, seed):
    """A smooth textured foreground, a soft-edged mask (3 equal channels like the alphamask JPEGs, plus per-channel noise so the
    channels are really read separately), a background and a source frame of another size."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(seed)
    tex = lambda h, w: np.clip(ndi.gaussian_filter(rng.normal(128, 90, (h, w, 3)), (1.2, 1.2, 0)), 0, 255).astype(np.uint8)  # noqa: E731
    yy, xx = np.mgrid[0:H, 0:W]
    d = np.hypot((yy - H * 0.55) / (H * 0.38), (xx - W * 0.45) / (W * 0.22))
    m = np.clip((1.05 - d) * 6, 0, 1) * 255
    m3 = np.clip(np.stack([m] * 3, -1) + rng.integers(-3, 4, (H, W, 3)), 0, 255).astype(np.uint8)
    fg = (tex(H, W) * (m3 / 255.)).astype(np.uint8)
    return fg, m3, tex(H, W), tex(sh, sw)


def _oracle_chain(o, fg, mask, bg, src, M, scale=1.2):
    H, W = fg.shape[:2]
    fs = o.rescale_fg(o.warp_affine(fg, M, H, W), scale)
    ms = o.rescale_fg(o.warp_affine(mask, M, H, W), scale)
    res = o.blend(fs, ms, bg)
    return res, np.concatenate((o.resize_linear(src, H, W), res), axis=1)


@pytest.mark.parametrize("H,W,sh,sw", [(120, 200, 150, 240), (135, 241, 135, 241), (67, 130, 40, 77), (3, 5, 3, 5), (16, 64, 32, 128), (17, 65, 9, 33),
                                      (1080, 1920, 1080, 1920), (1080, 1920, 720, 1280)])
def test_replace_compose_one_launch_bit_exact(oracle_lib, H, W, sh, sw):
    """usc_replace_compose_u8 == warp_affine -> rescale_fg -> blend (+ resize of the source frame) == the oracle chain, for the
    reference's translation matrices (sub-pixel shifts in float32), sizes whose 1.2-fold is not an integer (the cubic resize then maps
    coordinates with 1 / 1.2, not with old / new), widths that are not a multiple of 4 (byte stores) and a one-channel mask."""
    from video_unscreen_amd import _native as nv
    from video_unscreen_amd.pipeline_replace import ReplacePipeline
    fg, m3, bg, src = _inputs(H, W, sh, sw, 7 * H + W)
    pipe = ReplacePipeline(H, W, dev(bg), scale=1.2)
    assert np.array_equal(pipe.bg.cpu().numpy(), bg)                    # same size: cv2.resize copies
    for dx, dy in ((-0.2072, -2.6028), (-1.725405405405406, 8.128108108108108), (13.0, -7.5)):
        M = pipe.shift_matrix(dx, dy)
        res, cmp_img = pipe.compose(dev(fg), dev(m3), dev(src), M)
        res_c, cmp_c = pipe.compose_chain(dev(fg), dev(m3), dev(src), M)
        assert torch.equal(res, res_c) and torch.equal(cmp_img, cmp_c)
        want, want_cmp = _oracle_chain(oracle_lib, fg, m3, bg, src, M)
        assert np.array_equal(res.cpu().numpy(), want), (dx, dy)
        assert np.array_equal(cmp_img.cpu().numpy(), want_cmp)
        assert np.array_equal(want, oracle_lib.blend(oracle_lib.rescale_fg(oracle_lib.shift_fg(fg, dx, dy), 1.2),
                                                     oracle_lib.rescale_fg(oracle_lib.shift_fg(m3, dx, dy), 1.2), bg))
    assert H * W < 100 or 0.02 < (want != bg).mean()                    # the composite really changed pixels
    # one-channel mask, compare only / res only, another zoom, a general affine matrix
    m1 = np.ascontiguousarray(m3[..., 1])
    M = np.array([[0.98, 0.05, 3.25], [-0.04, 1.01, -2.5]], np.float64)
    for scale in (1.0, 1.1, 1.37):
        res, _ = nv.replace_compose(dev(fg), dev(m1), M, dev(bg), scale=scale)
        want, want_cmp = _oracle_chain(oracle_lib, fg, m1, bg, src, M, scale)
        assert np.array_equal(res.cpu().numpy(), want), scale
        none, cmp_img = nv.replace_compose(dev(fg), dev(m1), M, dev(bg), src=dev(src), scale=scale, want_res=False)
        assert none is None and np.array_equal(cmp_img.cpu().numpy(), want_cmp)
    with pytest.raises(nv.NativeError):
        nv.replace_compose(dev(fg), dev(m1), M, dev(bg), scale=0.9)      # a zoom < 1 is outside rescale_fg's use (window bound)


def test_replace_tool_full_size_raw_files(tmp_path, oracle_lib):
    """tools/replace/replace.py --io npy on 1920 x 1080 inputs (BASELINE.json configs[4] at its full size; raw arrays, so nothing is
    lossy): the correspondence mean it prints is the reference's comp_dx_dy flow (replace.py:28-41) on the oracle, res / compare files
    equal the oracle chain (replace.py:59-82) bit for bit."""
    H, W, n = 1080, 1920, 2
    root = tmp_path / "replace"
    src, tgt, bgd = root / "unscreen_img" / "test7", root / "unscreenbg_img" / "out7", root / "unscreen_img" / "bg"
    for d in (src, tgt, bgd):
        d.mkdir(parents=True)
    data = []
    for t in range(n):
        fg, m3, bg, frame = _inputs(H, W, 720, 1280, 100 + t)
        _, sm3, _, _ = _inputs(720, 1280, 8, 8, 200 + t)
        sm3 = np.roll(sm3, (9 * t + 4, -11), (0, 1))
        np.save(tgt / f"fg_{t:06d}.npy", fg)
        np.save(tgt / f"alphamask_{t:06d}.npy", m3)
        np.save(src / f"frame_{t:06d}.npy", frame)
        np.save(src / f"alphamask_{t:06d}.npy", sm3)
        data.append((fg, m3, frame, sm3))
    bg_small = _inputs(600, 1000, 8, 8, 5)[2]
    np.save(bgd / "test5_case.npy", bg_small)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replace", "replace.py"), "--src", "test7", "--data_root", str(root), "--io", "npy"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    dxs, dys = [], []
    for fg, m3, frame, sm3 in data:
        dc, sc = oracle_lib.get_center(m3), oracle_lib.get_center(oracle_lib.resize_linear(sm3, H, W))
        dxs.append(sc[0] - dc[0])
        dys.append(sc[1] - dc[1])
    dx, dy = float(np.mean(dxs)), float(np.mean(dys))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("Correspondence mean")][0]
    got_dx, got_dy = (float(v) for v in line.split(":")[1].split())
    assert abs(got_dx - dx) < 1e-9 and abs(got_dy - dy) < 1e-9 and (dx, dy) != (0.0, 0.0), (line, dx, dy)
    bg = oracle_lib.resize_linear(bg_small, H, W)
    M = np.float32([[1, 0, dx], [0, 1, dy]]).astype(np.float64)
    out = root / "merge_test_img" / "test7_out7"
    for t, (fg, m3, frame, _) in enumerate(data):
        want, want_cmp = _oracle_chain(oracle_lib, fg, m3, bg, frame, M)
        assert np.array_equal(np.load(out / f"res_{t:06d}.npy"), want), t
        assert np.array_equal(np.load(out / f"compare_{t:06d}.npy"), want_cmp), t


@pytest.mark.parametrize("H,W", [(67, 131), (3, 1), (120, 200), (1080, 1920)])
def test_gray_moments_pair_exact(oracle_lib, H, W):
    """usc_gray_moments_pair_u8 (both get_center calls of a frame in one launch) == two usc_gray_moments_u8 calls == exact integer
    moments of the oracle's RGB2GRAY image; and the centres the host loop derives equal the oracle's get_center."""
    from video_unscreen_amd import _native as nv
    rng = np.random.default_rng(H * W)
    a, b = (rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(2))
    out = torch.zeros(6, dtype=torch.int64, device="cuda")
    nv.gray_moments_pair(dev(a), dev(b), out)
    got = out.cpu().numpy().reshape(2, 3)
    for k, img in enumerate((a, b)):
        _, single = nv.gray_moments(dev(img), rgb_order=True)
        g = oracle_lib.bgr2gray(np.ascontiguousarray(img[..., ::-1])).astype(np.int64)
        yy, xx = np.mgrid[0:H, 0:W]
        want = np.array([g.sum(), (g * xx).sum(), (g * yy).sum()])
        assert np.array_equal(got[k], want) and np.array_equal(single.cpu().numpy(), want)
        cx, cy = int(want[1] / (want[0] + 1e-6)), int(want[2] / (want[0] + 1e-6))
        assert (cx, cy) == tuple(oracle_lib.get_center(img))


@pytest.mark.parametrize("H,W", [(136, 240), (61, 93)])
def test_replace_host_loop_equals_tool_flow(oracle_lib, H, W):
    """ReplaceHostLoop (pinned host -> pinned host, packed uploads, target masks kept in HBM between the passes) on 6 frames: the
    correspondence mean and every side-by-side image equal the oracle's comp_dx_dy + per-frame chain, for packed pairs and for
    (a, b) tuples with a source of another size."""
    from video_unscreen_amd.pipeline_replace import ReplaceHostLoop, ReplacePipeline, pack_pair
    n = 6                                            # (61 x 93: images of a packed stack start at odd byte addresses - the kernels' byte paths)
    for sh, sw in ((H, W), (90, 150)):
        frames = []
        for t in range(n):
            fg, m3, bg, fr = _inputs(H, W, sh, sw, 50 + t)
            sm3 = np.roll(_inputs(sh, sw, 8, 8, 70 + t)[1], (2 * t - 3, 5 - t), (0, 1))
            frames.append((fg, m3, fr, sm3))
        bg = _inputs(H, W, 8, 8, 9)[2]
        pipe = ReplacePipeline(H, W, dev(bg), max_frames=n)
        loop = ReplaceHostLoop(pipe, n, src_hw=(sh, sw), ring=2)
        packed = (sh, sw) == (H, W)
        t_ = torch.from_numpy
        masks_h = [pack_pair(t_(m3), t_(sm3)) if packed else (t_(m3).pin_memory(), t_(sm3).pin_memory()) for _, m3, _, sm3 in frames]
        inputs_h = [pack_pair(t_(fg), t_(fr)) if packed else (t_(fg).pin_memory(), t_(fr).pin_memory()) for fg, _, fr, _ in frames]
        sx, sy, cnt = loop.pass1(masks_h)
        dxs = [oracle_lib.get_center(oracle_lib.resize_linear(sm3, H, W))[0] - oracle_lib.get_center(m3)[0] for _, m3, _, sm3 in frames]
        dys = [oracle_lib.get_center(oracle_lib.resize_linear(sm3, H, W))[1] - oracle_lib.get_center(m3)[1] for _, m3, _, sm3 in frames]
        assert (sx, sy, cnt) == (float(sum(dxs)), float(sum(dys)), float(n))
        M = pipe.shift_matrix(sx / cnt, sy / cnt)
        compare_h = torch.zeros((n, H, 2 * W, 3), dtype=torch.uint8).pin_memory()
        loop.pass2(inputs_h, M, compare_h)
        torch.cuda.synchronize()
        for t, (fg, m3, fr, _) in enumerate(frames):
            assert np.array_equal(compare_h[t].numpy(), _oracle_chain(oracle_lib, fg, m3, bg, fr, M)[1]), (sh, sw, t)
        if packed:                                   # the whole clip as ONE tensor per pass: several frames per copy (6 frames, groups of 4: a ragged tail)
            from video_unscreen_amd.pipeline_replace import pack_clip
            loop4 = ReplaceHostLoop(ReplacePipeline(H, W, dev(bg), max_frames=n), n, ring=2, group=4)
            again = loop4.pass1(pack_clip([(t_(m3), t_(sm3)) for _, m3, _, sm3 in frames]))
            assert again == (sx, sy, cnt)
            cmp4 = torch.zeros((n, H, 2 * W, 3), dtype=torch.uint8).pin_memory()
            loop4.pass2(pack_clip([(t_(fg), t_(fr)) for fg, _, fr, _ in frames]), M, cmp4)
            torch.cuda.synchronize()
            assert torch.equal(cmp4, compare_h)


def test_replace_loop_matches_reference_python_golden():
    """The device loop against tests/golden/replace_loop.npz - outputs of the reference's OWN tools/replace/replace.py (comp_dx_dy and the
    frame loop lifted with ast, run over its own imgprocess.py on the cv2 shim: tests/golden/gen_replace.py): the correspondence mean and
    every res / compare image bit for bit, through ReplaceHostLoop (pinned host -> pinned host) with a source of another size."""
    from video_unscreen_amd.pipeline_replace import ReplaceHostLoop, ReplacePipeline
    z = np.load(os.path.join(ROOT, "tests", "golden", "replace_loop.npz"))
    n = sum(1 for k in z.files if k.startswith("res_"))
    H, W = z["tgt_fg_0"].shape[:2]
    sh, sw = z["src_frame_0"].shape[:2]
    pipe = ReplacePipeline(H, W, dev(z["bg"]), max_frames=n)
    loop = ReplaceHostLoop(pipe, n, src_hw=(sh, sw), ring=2)
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()          # noqa: E731
    sx, sy, cnt = loop.pass1([(t_(z[f"tgt_mask_{t}"]), t_(z[f"src_mask_{t}"])) for t in range(n)])
    assert [sx / cnt, sy / cnt] == z["dx_dy_mean"].tolist()
    compare_h = torch.zeros((n, H, 2 * W, 3), dtype=torch.uint8).pin_memory()
    loop.pass2([(t_(z[f"tgt_fg_{t}"]), t_(z[f"src_frame_{t}"])) for t in range(n)], pipe.shift_matrix(sx / cnt, sy / cnt), compare_h)
    torch.cuda.synchronize()
    for t in range(n):
        assert np.array_equal(compare_h[t].numpy(), z[f"compare_{t}"]), t
        assert np.array_equal(compare_h[t, :, W:].numpy(), z[f"res_{t}"]), t
