"""Experiment: run the shortcut branches on a second stream concurrently with the encoder."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet, ConvDesc

net = HipMattingNet(synthetic_state_dict(0), 544, 960)
lib = nv.load()
net.x8.normal_()
D = net._descs
n = len(D)
# layer indices: stem 0-2, enc 3-21, shortcuts 22-31, decoder 32..
enc = list(range(0, 22)); sc = list(range(22, 32)); dec = list(range(32, n))
def arr(idx): return (ConvDesc * len(idx))(*[D[i] for i in idx]), len(idx)
A_enc, A_sc, A_dec, A_all = arr(enc), arr(sc), arr(dec), arr(list(range(n)))
s1 = torch.cuda.current_stream(); s2 = torch.cuda.Stream()
def seq():
    lib.usc_conv_sequence_f32(A_all[0], A_all[1], s1.cuda_stream)
def two():
    ev0 = torch.cuda.Event(); ev0.record(s1)
    s2.wait_event(ev0)
    lib.usc_conv_sequence_f32(A_sc[0], A_sc[1], s2.cuda_stream)
    lib.usc_conv_sequence_f32(A_enc[0], A_enc[1], s1.cuda_stream)
    ev1 = torch.cuda.Event(); ev1.record(s2)
    s1.wait_event(ev1)
    lib.usc_conv_sequence_f32(A_dec[0], A_dec[1], s1.cuda_stream)
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
seq(); torch.cuda.synchronize(); ref = net.pred.clone()
two(); torch.cuda.synchronize(); print("two-stream output equal:", torch.equal(ref, net.pred))
# NOTE: sc0 reads x8 and encoder layer 0 reads x8 -> no hazard; shortcut of x1 needs layer 1 output -> the naive split
# above is only valid as a timing probe if sc1.. wait for their producers; here shortcuts start immediately (racy for
# correctness but same work), so compare time only.
for k in range(3):
    print("sequential %.3f ms   two-stream %.3f ms" % (timeit(seq), timeit(two)))
