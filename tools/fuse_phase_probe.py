"""Phase timing of the attention launch's weight-in-LDS form at the bench shape (a -DDN_FUSE_PHASES=1 build of libdisconet_hip.so:
AB_FILES=fuse_mlp tools/ab/build.sh DN_FUSE_PHASES 1; run with DISCONET_HIP_LIB=tools/ab/DN_FUSE_PHASES_1/libdisconet_hip.so):
mean cycles per active wave in staging + barriers / ego term / layer-1 passes of the list slots / tails / weighted sum."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disconet_amd import Config, DiscoNet, ops, _lib
from disconet_amd.synthetic import make_trans_matrices, randomize_bn_stats
lib = _lib.load()
fn = ctypes.CDLL(_lib.LIB_PATH).dn_fuse_phase_cycles
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
torch.manual_seed(0)
B, A, h, w, c = 4, 5, 32, 32, 256
model = DiscoNet(Config(map_hw=256), kd_flag=0, num_agent=A)
randomize_bn_stats(model)
model.eval().cuda()
P = model._get_plan()
feat = torch.randn(A * B, h, w, c, device="cuda")
trans = make_trans_matrices(B, A, jitter_seed=1).cuda()
na = torch.tensor([A] * B, dtype=torch.int32).cuda()
warped = torch.empty((B, A, A - 1, h, w, c), device="cuda")
ops.warp_neighbors(feat, trans, na, B, A, False, 0, A, out=warped, fm=True)
run = lambda: ops.disco_fuse_mlp(feat, warped, na, P["_fuse_mlp"], B, A, False, False, 0, A, sp_out=True, fm=True)
for _ in range(5):
    run()
buf = (ctypes.c_ulonglong * 8)()
fn(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
fn(buf, 1)
v = list(buf)
n = max(v[5], 1)
names = ["staging + barriers", "ego term (layer 1, W1_ego)", "layer 1 of the list slots", "tails (layers 2-4, exp)", "weighted sum + stores"]
print("launch %.1f us; %d active waves, %.0f cycles per wave (%.1f us at 2.4 GHz)" % (e0.elapsed_time(e1) * 1e3, n, v[6] / n, v[6] / n / 2400))
for k, name in enumerate(names):
    print("  %-28s %8.0f cycles  %5.1f %%" % (name, v[k] / n, 100.0 * v[k] / max(v[6], 1)))
print("  %-28s %8.0f cycles" % ("unaccounted", (v[6] - sum(v[:5])) / n))
