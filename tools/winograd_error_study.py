"""Error growth of Winograd F(2 x 2, 3 x 3) under the conv engine's split-f16 arithmetic, on the CPU (numpy): the number behind
DESIGN.md 3.1g's "on paper" paragraph (review item 2d of round 4: state the error growth against 1e-4 before writing a kernel).

Arithmetic model of the engine (csrc/conv_sp.hip): an operand is hi + lo with hi = half(x), lo = half(x - hi) (weights after a
power-of-two lift); a product is hi.hi + hi.lo + lo.hi (the lo.lo term is dropped), the sum over K runs in fp32.
  direct:    y = sum over (tap, channel) of the split products of x and w
  winograd:  V = B^T d B per 4 x 4 input tile (fp32 on hi + lo, then RE-SPLIT), U = G g G^T at pack time (fp32, lift, split),
             M = sum over channels of the split products U . V per tile position (fp32), y = A^T M A (fp32)
Both against the same conv in float64.  One 3x3 stride-1 layer, C_in = C_out = `c`, post-ReLU inputs, kaiming weights.

    python tools/winograd_error_study.py [c=256] [hw=32] [seeds=3]      -> profiles/r05_winograd_error.txt (stdout)"""
import sys

import numpy as np

B = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)        # B^T
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
A = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)                                       # A^T


def split(x):
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def lift_of(w):
    return 2.0 ** (12 - np.floor(np.log2(np.abs(w).max())))


def split_dot(xh, xl, wh, wl, spec):
    """sum over the contracted axes of hi.hi + hi.lo + lo.hi, each partial sum and the total in fp32"""
    e = lambda a, b: np.einsum(spec, a, b, dtype=np.float32, optimize=True)
    return (e(xh, wh) + e(xh, wl)) + e(xl, wh)


def main(c=256, hw=32, seeds=3):
    rows = []
    for seed in range(seeds):
        rng = np.random.default_rng(seed)
        x = np.maximum(rng.standard_normal((hw + 2, hw + 2, c)), 0.0)            # one padded map, post-ReLU
        x[0], x[-1], x[:, 0], x[:, -1] = 0, 0, 0, 0
        w = rng.standard_normal((3, 3, c, c)) * np.sqrt(2.0 / (9 * c))           # [ky][kx][ci][co], kaiming
        ref = sum(np.einsum("hwi,io->hwo", x[ky:ky + hw, kx:kx + hw], w[ky, kx]) for ky in range(3) for kx in range(3))
        # --- direct
        xh, xl = split(x)
        wm = lift_of(w)
        wh, wl = split(w * wm)
        y = np.zeros((hw, hw, c), np.float32)
        for ky in range(3):
            for kx in range(3):
                y += split_dot(xh[ky:ky + hw, kx:kx + hw], xl[ky:ky + hw, kx:kx + hw], wh[ky, kx], wl[ky, kx], "hwi,io->hwo")
        y_direct = y.astype(np.float64) / wm
        # --- winograd F(2 x 2, 3 x 3): tiles of 4 x 4 inputs at stride 2
        t = hw // 2
        xs = (xh + xl).astype(np.float32)                                       # the engine transforms what it stores: hi + lo
        d = np.stack([np.stack([xs[2 * i:2 * i + 4, 2 * j:2 * j + 4] for j in range(t)]) for i in range(t)])     # [ti][tj][4][4][c]
        V = np.einsum("ab,ijbdc,ed->ijaec", B.astype(np.float32), d, B.astype(np.float32), dtype=np.float32)
        U = np.einsum("ak,klio,bl->abio", G, w, G)                               # [4][4][ci][co], float64 at pack time
        um = lift_of(U)
        uh, ul = split((U * um).astype(np.float32))
        vh, vl = split(V)
        M = split_dot(vh, vl, uh, ul, "ijabc,abco->ijabo")
        yw = np.einsum("pa,ijabo,qb->ijpqo", A.astype(np.float32), M, A.astype(np.float32), dtype=np.float32)
        y_wino = yw.transpose(0, 2, 1, 3, 4).reshape(hw, hw, c).astype(np.float64) / um
        scale = np.abs(ref).max()
        rows.append((seed, scale, np.abs(y_direct - ref).max(), np.abs(y_wino - ref).max(), np.abs(V).max() / np.abs(xs).max(),
                     np.sqrt(np.mean((y_direct - ref) ** 2)), np.sqrt(np.mean((y_wino - ref) ** 2))))
    print("Winograd F(2x2, 3x3) against the direct form under the split-f16 x3 arithmetic (numpy model; tools/winograd_error_study.py)")
    print("layer: 3x3 stride 1, %d -> %d channels, %d x %d map, post-ReLU N(0,1) inputs, kaiming weights; reference: float64" % (c, c, hw, hw))
    print("%4s %10s %14s %14s %10s %12s %12s" % ("seed", "max |y|", "direct max err", "winograd max", "|V|/|x|", "direct rms", "winograd rms"))
    for r in rows:
        print("%4d %10.3f %14.3e %14.3e %10.2f %12.3e %12.3e" % r)
    md, mw = max(r[2] for r in rows), max(r[3] for r in rows)
    print("worst case over the seeds: direct %.2e, winograd %.2e (%.1f x) on outputs of magnitude ~%.1f; the whole-model bar is 1e-4 on O(1)-O(5) logits"
          % (md, mw, mw / md, np.mean([r[1] for r in rows])))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    main(*a)
