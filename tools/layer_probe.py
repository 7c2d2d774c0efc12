"""conv1_1 (32 -> 64, stride 2, 256^2) runs 80 us inside the step and 61 us on random data in sp_conv_check:
which property of the real input makes the difference?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disconet_amd import Config, DiscoNet, ops
from disconet_amd.synthetic import make_sparse_scene_batch, randomize_bn_stats
torch.manual_seed(0)
A, B, HW = 5, 4, 256
model = DiscoNet(Config(map_hw=HW), kd_flag=0, num_agent=A)
randomize_bn_stats(model)
model.eval().cuda()
P = model._get_plan()
indices, offsets, _ = make_sparse_scene_batch(B, A, HW)
bevs = ops.scatter_dense_sp(indices.cuda(), offsets.cuda(), A * B, (HW, HW, 13))
x0 = P["conv_pre_2"].run(P["conv_pre_1"].run(bevs))
L = P["conv1_1"]
def t(x, tag):
    L.run(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.run(x)
    e1.record(); torch.cuda.synchronize()
    d = x.data.float()
    print("%-34s %.1f us   zeros %.1f %%  |x| max %.2f  ptr %% 2MiB = %d KiB" % (tag, 50 * e0.elapsed_time(e1), 100 * float((d == 0).float().mean()), float(d.abs().max()), (x.data.data_ptr() % (1 << 21)) >> 10))
t(x0, "real x0 (conv_pre_2 output)")
xr = ops.SpTensor.from_nhwc(torch.randn(A * B, HW, HW, 32, device="cuda"))
t(xr, "random normal")
t(ops.SpTensor.from_nhwc(torch.relu(torch.randn(A * B, HW, HW, 32, device="cuda"))), "relu(random normal)")
xc = ops.SpTensor(A * B, HW, HW, 32, device="cuda", data=x0.data.clone())
t(xc, "clone of the real x0")
t(ops.SpTensor.from_nhwc(x0.nhwc() * 0 + 1.0), "all ones")
t(ops.SpTensor.from_nhwc(x0.nhwc() * 0), "all zeros")
for k in (4, 16, 64, 1024):
    t(ops.SpTensor.from_nhwc(x0.nhwc() * float(k)), "real x0 x %d" % k)
t(ops.SpTensor.from_nhwc(torch.randn(A * B, HW, HW, 32, device="cuda") * 0.05), "random normal x 0.05")
