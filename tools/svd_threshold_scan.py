"""f32 emulation of the device Jacobi SVD (sparkl_amd/csrc/device_math.h): singular-value error and sweep count versus the
stop threshold on the squared relative column coupling, for F = I + scale * N(0,1).  Motivates the 1e-12 in jacobi_pair."""
import numpy as np
f = np.float32


def pair(W, V, p, q, tau):
    wp, wq, vp, vq = W[:, p].copy(), W[:, q].copy(), V[:, p].copy(), V[:, q].copy()
    a, b, g = f(wp @ wp), f(wq @ wq), f(wp @ wq)
    g2, ab = f(g * g), f(a * b)
    c, s = f(1), f(0)
    if g2 > f(1e-18) * ab:
        z = f((b - a) / (f(2) * g))
        t = f(np.copysign(1, z)) / f(abs(z) + np.sqrt(f(1) + z * z))
        c = f(1 / np.sqrt(f(1) + t * t))
        s = f(c * t)
    W[:, p], W[:, q], V[:, p], V[:, q] = c * wp - s * wq, s * wp + c * wq, c * vp - s * vq, s * vp + c * vq
    return g2 > f(tau) * ab


def svd(A, tau):
    W, V, n = A.astype(f).copy(), np.eye(3, dtype=f), 0
    for _ in range(8):
        n += 1
        m = pair(W, V, 0, 1, tau)
        m |= pair(W, V, 0, 2, tau)
        m |= pair(W, V, 1, 2, tau)
        if not m:
            break
    return np.linalg.norm(W, axis=0).astype(f), n


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    mats = {sc: [np.eye(3) + sc * rng.normal(size=(3, 3)) for _ in range(1500)] for sc in (1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.05, 0.2)}
    for tau in (1e-7, 1e-9, 1e-10, 1e-11, 1e-12, 1e-13):
        out = []
        for sc, ms in mats.items():
            r = [svd(A, tau) for A in ms]
            err = max(np.abs(np.sort(S) - np.sort(np.linalg.svd(A, compute_uv=False))).max() for (S, _), A in zip(r, ms))
            out.append(f"{sc:g}: err {err:.1e} sweeps {np.mean([n for _, n in r]):.2f}")
        print(f"tau {tau:g} |", " | ".join(out))
