"""Worst-case magnitude of the wide parts (L, H) in the paired internal rounds of poseidon2_fast.cuh (p2f_permute): inputs and
S-box outputs have parts < 2^32; p2f_fold_signed needs |part| < 2^61.  Prints log2 of the largest part after every round for three
refold schedules; the kernel refolds after rounds 3, 7, 11, 15, 19 (first line).
Usage: python tools/p2_pair_bounds.py"""
import math
B32 = 2**32
def run(refold_after):
    # round 0
    x = B32  # parts of inputs
    y = B32
    R = 11 * x
    S = 2 * R + y
    s8 = 8 * S
    X1 = 16 * x + s8; X2 = 32 * x + s8; X11 = 2 * x + s8
    k8 = [4, 24, 32, 2]
    hprev = [2 * x] * 4                      # h(0)
    hcur = [k * x + s8 + k * x for k in k8]  # h(1)
    t0in = s8 + 16 * y + B32
    mx = max([X1, X2, X11, t0in] + hcur + hprev)
    worst = mx
    for r in range(1, 22):
        hs = sum(hcur)
        R = 2 * hs + X1 + X2 + X11
        S = R + y
        s8 = 8 * S
        t0in = s8 + 16 * y + B32
        X1 = 8 * X1 + s8; X2 = 16 * X2 + s8; X11 = X11 + s8
        hnext = [k * k * hp + s8 for k, hp in zip(k8, hprev)]
        # intermediate of 576: (h*9) << 6 same magnitude
        hprev, hcur = hcur, hnext
        mx = max([X1, X2, X11, t0in] + hcur + hprev)
        worst = max(worst, mx)
        print(r, round(math.log2(mx), 2), end=' | ')
        if r in refold_after:
            X1 = X2 = X11 = B32
            hprev = [B32] * 4; hcur = [B32] * 4
    # final recombination magnitude
    fin = max(hc + k * hp for k, hc, hp in zip(k8, hcur, hprev))
    print('\nfinal', round(math.log2(fin), 2), 'worst', round(math.log2(worst), 2))
run({3, 7, 11, 15, 19})
run({2, 6, 10, 14, 18})
run({2, 5, 8, 11, 14, 17, 20})
