"""CPU check behind DESIGN 9's config-5 lead (no GPU, no library): does a 4-wide block iteration with its blocks V / W STORED
in fp32 (sums in fp64 registers, as a kernel would do), closed by ONE step in fp64, reach the 1e-10 score gate on 12-taxon
splits?  Same iteration as k_sparse_slow: alternating half products W = C^T V, Y = C W, Cholesky-QR of every fresh block
(fp64 4 x 4 Gram), Ritz sum = trace of the Gram matrix of the fresh block.
    python tools/experiments/mixed_precision_check.py [n_taxa] [n_sites] [seed] [branch length]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from splitp_amd import synthetic as syn
from oracle import splitp_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
branch = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
keys, counts = syn.pattern_table(syn.simulate_sites(n, L, branch, seed=seed))
rng = np.random.default_rng(seed)
print(f"{n} taxa, {L} sites, branch {branch}, {len(keys)} patterns")


def orth(X):
    """Cholesky-QR in fp64 (the kernel's 4 x 4 Gram + factor); returns the orthonormal block and the Gram matrix's trace."""
    S = X.T @ X
    Lc = np.linalg.cholesky(S)
    return np.linalg.solve(Lc, X.T).T, float(np.trace(S))


def iterate(M, halves32, store32=True):
    """halves32 half products with the blocks rounded to fp32 after every orthonormalisation (store32), then one closing
    pair of half products in fp64.  Returns the score from the closing Ritz sum."""
    R, Cc = M.shape
    trace = float(np.sum(M * M))
    # start block: unit vectors on the rows of the 4 largest row sums (the kernel starts from the most frequent patterns)
    V = np.zeros((R, 4)); top = np.argsort(-M.sum(axis=1))[:4]; V[top, np.arange(4)] = 1.0
    V += 1e-3 * rng.standard_normal(V.shape); V, _ = orth(V)
    X, odd = V, True
    r32 = (lambda A: A.astype(np.float32).astype(np.float64)) if store32 else (lambda A: A)
    X = r32(X)
    for h in range(halves32):
        Y = (M.T @ X) if odd else (M @ X)        # sums in fp64, inputs as stored
        X, _ = orth(Y)
        X = r32(X)
        odd = not odd
    # closing step in fp64: re-orthonormalise the stored block, then two half products; Ritz sum of sigma^2 = trace of the
    # Gram matrix of (C^T V) for orthonormal V (odd) / of (C W) for orthonormal W (even)
    X, _ = orth(X)
    Y = (M.T @ X) if odd else (M @ X)
    X2, s1 = orth(Y)
    Y2 = (M @ X2) if odd else (M.T @ X2)
    _, s2 = orth(Y2)
    return np.sqrt(max(0.0, 1.0 - s2 / trace)), np.sqrt(max(0.0, 1.0 - s1 / trace))


names = list(range(n))
worst = {}
t0 = time.time()
for k in (n // 2, n // 2 - 1, 4):
    for trial in range(6):
        left = sorted(rng.choice(n, size=k, replace=False).tolist())
        right = [t for t in range(n) if t not in left]
        M = O.reduced_flattening_packed(keys, counts.astype(np.float64) / float(counts.sum()), n, left, right)[0]
        sv = np.linalg.svd(M, compute_uv=False)
        exact = np.sqrt(max(0.0, 1.0 - float(np.sum(sv[:4] ** 2)) / float(np.sum(sv ** 2))))
        row = [f"{k}|{n - k} {M.shape} s4/s5 {sv[3] / sv[4]:.3f} score {exact:.6f}"]
        for halves in (4, 6, 8):
            s64, _ = iterate(M, halves, store32=False)
            s32, s32_one = iterate(M, halves, store32=True)
            row.append(f"h={halves}: fp64 {abs(s64 - exact):.1e} fp32+close {abs(s32 - exact):.1e} (one half product only {abs(s32_one - exact):.1e})")
            worst[halves] = max(worst.get(halves, 0.0), abs(s32 - exact))
        print("  ".join(row))
print("worst |score - exact| with fp32 blocks + one fp64 closing pair:", {h: f"{v:.1e}" for h, v in worst.items()}, "%.0f s" % (time.time() - t0))
