"""Golden vectors for ISegAgent's no-BRS forward: the reference's OWN Python, run in the build container -
unscreen/iseg/agent.py (ISegAgent: preprocess, forward, postprocess), iseg/brs/brs_predictor.py (BasePredictor: points tensor, flip
test-time augmentation, sigmoid), iseg/brs/transforms.py, clicker.py and the model files iseg/brs/models/{distmap, deeplab_v3,
resnet, resnetv1b, basic_blocks}.py - loaded by path into the stand-in `unscreen` package of gen_orchestration.py (`cv2` = the
oracle-backed shim: pad_resize / imnormalize / inv_pad_resize run as the reference wrote them), on keyed-deterministic weights
(video_unscreen_amd/iseg/net.py: iseg_synthetic_state_dict; no loader that fetches is touched: the model is built with
pretrained_base=False as iseg/agent.py:29-33 does).

One fact about the reference, recorded in the fixture: with with_brs=False the agent builds a `BasePredictor`, whose
get_prediction() first calls self._get_clicks_maps(clicker) - abstract in BasePredictor (brs_predictor.py:104-105 raises
NotImplementedError) - so ISegAgent.forward raises before it reaches the network.  The click maps are not used by
BasePredictor._get_prediction (:53-56); the fixture is produced with that one method bound to the reference's own implementation
of it (BRSBasePredictor._get_clicks_maps, :126-128), everything else as it stands.

Output: tests/golden/iseg_90x160.npz (data only).  Usage: python tests/golden/gen_iseg.py"""
import os
import sys
import tempfile
import types
import zlib

import scipy.optimize  # noqa: F401  (before gen_orchestration restores np.bool = bool for the reference: numpy.ma must import first)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_orchestration as go      # noqa: E402

REF = go.REF


def load_iseg():
    import importlib.util
    utils, _, _, shim = go.load_reference()

    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    pkg("unscreen.iseg")
    pkg("unscreen.iseg.brs")
    pkg("unscreen.iseg.brs.models")
    for f in ("basic_blocks", "resnetv1b", "resnet", "deeplab_v3", "distmap"):
        load(f"unscreen.iseg.brs.models.{f}", f"unscreen/iseg/brs/models/{f}.py")
    for f in ("clicker", "misc", "transforms", "brs_optimizer", "brs_predictor"):
        load(f"unscreen.iseg.brs.{f}", f"unscreen/iseg/brs/{f}.py")
    brs = sys.modules["unscreen.iseg.brs"]
    for src, names in (("brs_optimizer", ("ScaleBiasOptimizer",)), ("brs_predictor", ("BasePredictor", "FeatureBRSPredictor")), ("clicker", ("Click", "Clicker"))):
        for n in names:
            setattr(brs, n, getattr(sys.modules[f"unscreen.iseg.brs.{src}"], n))
    brs.SepConvHead = sys.modules["unscreen.iseg.brs.models.basic_blocks"].SepConvHead
    brs.DeepLabV3Plus = sys.modules["unscreen.iseg.brs.models.deeplab_v3"].DeepLabV3Plus
    brs.DistMapsModel = sys.modules["unscreen.iseg.brs.models.distmap"].DistMapsModel
    agent = load("unscreen.iseg.agent", "unscreen/iseg/agent.py")
    return agent.ISegAgent, sys.modules["unscreen.iseg.brs.brs_predictor"], shim


def main():
    sys.path.insert(0, go.ROOT)
    from video_unscreen_amd.iseg.net import iseg_checkpoint_layout, iseg_synthetic_state_dict
    from video_unscreen_amd.synth import SyntheticClip
    ISegAgent, pred_mod, shim = load_iseg()
    sd = iseg_synthetic_state_dict(0)
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "iseg_synth.pth")
        torch.save(sd, ck)
        agent = ISegAgent(ck, with_brs=False, input_long_side=128, prob_thresh=0.5, with_flip=True, cuda_device=-1)
    model = agent.predictor.net
    rsd = model.state_dict()
    lay = iseg_checkpoint_layout()
    assert list(rsd.keys()) == [k for k, _ in lay], "state-dict key list differs from the reference's"
    assert all(tuple(rsd[k].shape) == tuple(sh) for k, sh in lay)
    assert all(torch.equal(rsd[k], sd[k]) for k, _ in lay), "strict=False must still have loaded every tensor"
    assert type(agent.predictor).__name__ == "BasePredictor"

    H, W = 90, 160
    frame = SyntheticClip(H, W, 1, config_id=2, device="cpu").frame(0)["frame"].numpy()
    clicks = [(True, 40, 70), (True, 62, 101), (False, 9, 21)]                 # (is_positive, y, x) in frame coordinates (agent.py:98-101)
    raised = 0
    try:
        agent.forward(frame.copy(), clicks)
    except NotImplementedError:
        raised = 1                                                             # brs_predictor.py:104-105 (see the module docstring)
    agent.predictor._get_clicks_maps = types.MethodType(pred_mod.BRSBasePredictor._get_clicks_maps, agent.predictor)

    cap = {}
    orig_forward = model.forward

    def spy(image, points):
        cap["image"], cap["points"] = image.clone(), points.clone()
        coord = model.dist_maps(image, points)
        cap["coord"] = coord.clone()
        cap["x"] = model.rgb_conv(torch.cat((image, coord), dim=1)).clone()
        out = orig_forward(image, points)
        cap["logits"] = out.clone()
        return out

    model.forward = spy
    arg = frame.copy()
    with torch.no_grad():
        img_tensor, clicker = agent.preprocess(arg.copy(), clicks)
        mask = agent.forward(arg, clicks)
        agent.predictor.set_input_image(img_tensor)
        prob = agent.predictor.get_prediction(clicker)
    assert np.array_equal(arg, frame), "ISegAgent.forward must not modify its frame"
    assert np.array_equal(mask, agent.postprocess(prob, (H, W)))
    out = {"frame": frame, "clicks": np.array([[int(p), y, x] for p, y, x in clicks], np.int64), "input_long_side": np.int64(128),
           "img_tensor": img_tensor.numpy(), "points": cap["points"].numpy(), "coord": cap["coord"].numpy(), "x": cap["x"].numpy(),
           "logits": cap["logits"].numpy(), "prob": prob, "mask": mask, "ref_nobrs_raises_notimplemented": np.int64(raised),
           "n_keys": np.int64(len(lay)), "keys_crc": np.int64(sum(zlib.crc32((k + str(tuple(sh))).encode()) for k, sh in lay))}
    # a second case: no flip, a single negative click only (the positive slot is all (-1, -1): the 1e6 branch of DistMaps), portrait
    agent2_frame = np.ascontiguousarray(frame.transpose(1, 0, 2))
    agent.with_flip = False
    agent.predictor.with_flip = False
    agent.predictor.transforms = agent.predictor.transforms[:1]
    clicks2 = [(False, 80, 45)]
    with torch.no_grad():
        mask2 = agent.forward(agent2_frame.copy(), clicks2)
    out.update({"frame2": agent2_frame, "clicks2": np.array([[0, 80, 45]], np.int64), "points2": cap["points"].numpy(), "logits2": cap["logits"].numpy(),
                "coord2": cap["coord"].numpy(), "mask2": mask2})
    np.savez_compressed(os.path.join(HERE, "iseg_90x160.npz"), **out)
    print("iseg: reference no-BRS forward raises NotImplementedError:", bool(raised), "| input", tuple(img_tensor.shape), "points", cap["points"].shape,
          "| prob range %.3f..%.3f" % (prob.min(), prob.max()), "| mask fg %.3f, mask2 fg %.3f" % ((mask > 0).mean(), (mask2 > 0).mean()),
          "| cv2 calls:", sorted({c[0] for c in shim.calls}))


if __name__ == "__main__":
    main()
