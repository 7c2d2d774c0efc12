"""Why the split-f16 gradients need a power-of-two lift, on the CPU (numpy): the number behind DESIGN.md 8's "the 3.6 % of rounds 2-4
was the f16 RANGE, not the 22-bit significand".

A weight-gradient entry is dW = sum over P pixels of dz[p] * x[p].  Operands enter the f16 MFMA as hi + lo (hi = half(v),
lo = half(v - hi)), a product is hi.hi + hi.lo + lo.hi, the sum runs in fp32.  A gradient map of magnitude 1e-4 sits at the bottom of
binary16's normal range (2^-14 = 6.1e-5): its lo halves are subnormals with a 6e-8 grid -- nearly all of them round to 0 or one grid
step, so the pair carries ~11 bits, not 22.  Multiplied by 2^k with max |dz| * 2^k ~ 2^8 first (and 1 / 2^k folded into an exact scale
afterwards) the same pair carries its 22 bits for 19 binades below the maximum.

    python tools/lift_error_study.py      -> profiles/r05_lift_error.txt (stdout)"""
import numpy as np


def split(v):
    v = v.astype(np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def split_sum(dz, x, lift):
    dh, dl = split(dz * lift)
    xh, xl = split(x * 16.0)                      # the activations' fixed lift (train.py :: _WGRAD_X_LIFT)
    prod = (dh * xh + dh * xl) + dl * xh          # each product exact in fp32's 24 bits to first order; summed below in fp32 blocks
    acc = np.float32(0)
    for blk in np.array_split(prod.astype(np.float32), 1024):
        acc = np.float32(acc + np.float32(blk.sum(dtype=np.float32)))
    return float(acc) / (lift * 16.0)


def main():
    rng = np.random.default_rng(0)
    P = 1 << 20                                    # ~ the 1.3 M pixels of a 256 x 256 layer at batch 4 x 5 agents
    print("dW = sum over %d pixels of dz * x: relative error |dW - dW64| / (rms of a term * sqrt(P)) -- the scale a gradient entry has" % P)
    print("x: post-ReLU N(0, 1); dz: N(0, s) with a log-normal envelope (heavy tail, max ~30 x the rms), as the maps of the training step")
    print("%10s %12s %14s %14s %14s %16s" % ("rms(dz)", "lift 2^k", "fp32 chain", "split, no lift", "split, lifted", "bits, no lift"))
    for s in (1e-2, 1e-3, 1e-4, 1e-5):
        errs = []
        for trial in range(8):
            x = np.maximum(rng.standard_normal(P), 0.0)
            dz = rng.standard_normal(P) * s * np.exp(0.8 * rng.standard_normal(P))
            ref = float(np.dot(dz, x))
            scale = float(np.sqrt(np.mean((dz * x) ** 2)) * np.sqrt(P))
            f32 = float(np.dot(dz.astype(np.float32), x.astype(np.float32)))
            k = 8 - int(np.floor(np.log2(np.abs(dz).max())))
            dh, dl = split(dz)
            nz = dz != 0
            sub = float(np.median(-np.log2(np.maximum(np.abs(dz[nz] - (dh + dl)[nz]) / np.abs(dz[nz]), 2.0 ** -30))))     # bits the un-lifted pair keeps
            errs.append((abs(f32 - ref) / scale, abs(split_sum(dz, x, 1.0) - ref) / scale, abs(split_sum(dz, x, 2.0 ** k) - ref) / scale, sub, k))
        e = np.array([r[:4] for r in errs])
        print("%10.0e %12s %14.2e %14.2e %14.2e %16.1f" % (s, "2^%d" % errs[0][4], e[:, 0].mean(), e[:, 1].mean(), e[:, 2].mean(), e[:, 3].mean()))
    print("(bits, no lift: median significant bits of an un-lifted hi + lo pair; means over 8 draws; the backward of a BatchNorm then amplifies a relative error of its input gradient by ~1e2-1e4: 1e-4 here is the")
    print(" percent-level error rounds 2-4 measured on conv5_1.weight without a lift, 1e-7 is the fp32 kernels' own level)")


if __name__ == "__main__":
    main()
