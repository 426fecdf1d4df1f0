"""Plain-text .g2o reader/writer and the synthetic SE3 generator (host-side data plumbing).

Format (reference datasets/*.g2o; parsed by g2o in the reference, src/graph_wrapper_g2o.cpp:107-154):
    VERTEX_SE2 id x y theta                 EDGE_SE2 i j dx dy dtheta + 6 upper-triangular info
    VERTEX_SE3:QUAT id x y z qx qy qz qw    EDGE_SE3:QUAT i j (7) + 21 upper-triangular info
Quaternions are normalised on load (g2o normalises edge measurements; vertices are normalised too so
that every rotation the kernels see is orthonormal — stated in DESIGN.md).
The native reader in csrc/host (spg_graph_load_g2o) follows the same rules; tests compare the two.
"""
import numpy as np


def load_g2o(path):
    vid, vpose, eij, edata = [], [], [], []
    d = None
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            tag = t[0]
            if tag == "VERTEX_SE2":
                d = 3 if d is None else d
                vid.append(int(t[1]))
                vpose.append([float(x) for x in t[2:5]])
            elif tag == "EDGE_SE2":
                eij.append((int(t[1]), int(t[2])))
                edata.append([float(x) for x in t[3:12]])
            elif tag == "VERTEX_SE3:QUAT":
                d = 6 if d is None else d
                vid.append(int(t[1]))
                vpose.append([float(x) for x in t[2:9]])
            elif tag == "EDGE_SE3:QUAT":
                eij.append((int(t[1]), int(t[2])))
                edata.append([float(x) for x in t[3:31]])
    ids = np.asarray(vid, np.int32)
    poses = np.asarray(vpose, np.float64)
    order = np.argsort(ids, kind="stable")
    ids, poses = ids[order], poses[order]
    edge_ij = np.asarray(eij, np.int32).reshape(-1, 2)
    edge_data = np.asarray(edata, np.float64)
    if d == 6:
        poses[:, 3:7] /= np.linalg.norm(poses[:, 3:7], axis=1, keepdims=True)
        if len(edge_data):
            edge_data[:, 3:7] /= np.linalg.norm(edge_data[:, 3:7], axis=1, keepdims=True)
    return {"pose_dim": d, "ids": ids, "poses": poses, "edge_ij": edge_ij, "edge_data": edge_data}


def write_g2o(path, g):
    d = g["pose_dim"]
    vt, et = ("VERTEX_SE2", "EDGE_SE2") if d == 3 else ("VERTEX_SE3:QUAT", "EDGE_SE3:QUAT")
    with open(path, "w") as f:
        for i, p in zip(g["ids"], g["poses"]):
            f.write(f"{vt} {int(i)} " + " ".join(repr(float(x)) for x in p) + "\n")
        for (a, b), r in zip(g["edge_ij"], g["edge_data"]):
            f.write(f"{et} {int(a)} {int(b)} " + " ".join(repr(float(x)) for x in r) + "\n")


# ------------------------------------------------------------------ quaternion helpers (x y z w)
def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def quat_conj(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def quat_rotate(q, v):
    qv = np.concatenate([v, np.zeros(v.shape[:-1] + (1,))], axis=-1)
    return quat_mul(quat_mul(q, qv), quat_conj(q))[..., :3]


def quat_from_R(R):
    """Batch rotation matrices (…,3,3) -> unit quaternions with w >= 0 (branch on the largest pivot)."""
    R = np.asarray(R, np.float64)
    q = np.empty(R.shape[:-2] + (4,))
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    cand = np.stack([1 + R[..., 0, 0] - R[..., 1, 1] - R[..., 2, 2],
                     1 - R[..., 0, 0] + R[..., 1, 1] - R[..., 2, 2],
                     1 - R[..., 0, 0] - R[..., 1, 1] + R[..., 2, 2], 1 + tr], axis=-1)
    which = np.argmax(cand, axis=-1)
    for c in range(4):
        m = which == c
        if not np.any(m):
            continue
        Rm = R[m]
        s = 2.0 * np.sqrt(cand[m][:, c])
        if c == 3:
            qq = np.stack([(Rm[:, 2, 1] - Rm[:, 1, 2]) / s, (Rm[:, 0, 2] - Rm[:, 2, 0]) / s,
                           (Rm[:, 1, 0] - Rm[:, 0, 1]) / s, 0.25 * s], axis=-1)
        else:
            i, j, k = c, (c + 1) % 3, (c + 2) % 3
            qq = np.empty((Rm.shape[0], 4))
            qq[:, i] = 0.25 * s
            qq[:, j] = (Rm[:, j, i] + Rm[:, i, j]) / s
            qq[:, k] = (Rm[:, k, i] + Rm[:, i, k]) / s
            qq[:, 3] = (Rm[:, k, j] - Rm[:, j, k]) / s
        q[m] = qq
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    q[q[..., 3] < 0] *= -1
    return q


# ------------------------------------------------------------------ counter-based PRNG
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter, seed):
    """SplitMix64 output for stream positions `counter` (uint64 array) of the given seed."""
    with np.errstate(over="ignore"):
        z = (np.asarray(counter, np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def normal_stream(n, seed, stream=0):
    """n standard normals: Box-Muller on pairs of SplitMix64 uniforms (53-bit mantissas)."""
    m = (n + 1) // 2
    base = np.uint64(stream) * np.uint64(1 << 40)
    c = np.arange(2 * m, dtype=np.uint64) + base
    u = (splitmix64(c, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    u1 = 1.0 - u[0::2]  # (0,1]
    u2 = u[1::2]
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.empty(2 * m)
    z[0::2] = r * np.cos(2 * np.pi * u2)
    z[1::2] = r * np.sin(2 * np.pi * u2)
    return z[:n]


def synth_sphere(n_poses=100000, ring=400, radius=50.0, seed=20240611,
                 info_diag=(10.0, 10.0, 10.0, 400.0, 400.0, 100.0)):
    """Synthetic SE3 pose graph: sphere.g2o's structure scaled up.

    n_poses poses on a pole-to-pole spiral of `ring` poses per turn on a sphere; odometry edges
    (i, i+1) and ring-closure edges (i, i+ring). Vertex estimates are the ground truth (the
    linearisation point is an input of the path); measurement = true relative pose composed with
    (dt, dq) noise ~ N(0, diag(info)^-1); information = diag(info) written as 21 upper-tri numbers.
    """
    N, L = int(n_poses), int(ring)
    i = np.arange(N, dtype=np.float64)
    phi = np.pi * (i + 0.5) / N                    # polar angle, pole to pole
    th = 2 * np.pi * i / L                         # azimuth
    pos = radius * np.stack([np.sin(phi) * np.cos(th), np.sin(phi) * np.sin(th), np.cos(phi)], axis=-1)
    up = pos / np.linalg.norm(pos, axis=-1, keepdims=True)          # z axis: radial
    tang = np.stack([-np.sin(th), np.cos(th), np.zeros(N)], axis=-1)  # x axis: along the ring
    yax = np.cross(up, tang)
    yax /= np.linalg.norm(yax, axis=-1, keepdims=True)
    xax = np.cross(yax, up)
    R = np.stack([xax, yax, up], axis=-1)          # columns = body axes in the world frame
    q = quat_from_R(R)
    poses = np.concatenate([pos, q], axis=-1)
    a = np.concatenate([np.arange(N - 1), np.arange(N - L)]).astype(np.int32)
    b = np.concatenate([np.arange(1, N), np.arange(L, N)]).astype(np.int32)
    E = len(a)
    qa, qb = q[a], q[b]
    zt = quat_rotate(quat_conj(qa), pos[b] - pos[a])
    zq = quat_mul(quat_conj(qa), qb)
    sig = 1.0 / np.sqrt(np.asarray(info_diag))
    noise = normal_stream(6 * E, seed).reshape(E, 6) * sig
    dq = np.concatenate([noise[:, 3:], np.sqrt(np.maximum(0.0, 1 - np.sum(noise[:, 3:] ** 2, axis=1, keepdims=True)))], axis=-1)
    mt = zt + quat_rotate(zq, noise[:, :3])
    mq = quat_mul(zq, dq)
    mq /= np.linalg.norm(mq, axis=-1, keepdims=True)
    mq[mq[:, 3] < 0] *= -1
    info = np.zeros((E, 21))
    p = 0
    for r_ in range(6):
        for c_ in range(r_, 6):
            if r_ == c_:
                info[:, p] = info_diag[r_]
            p += 1
    edge_data = np.concatenate([mt, mq, info], axis=-1)
    return {"pose_dim": 6, "ids": np.arange(N, dtype=np.int32), "poses": poses,
            "edge_ij": np.stack([a, b], axis=-1), "edge_data": edge_data}


def synth_manhattan(n_poses=10000, side=100, seed=7, info_diag=(44.7214, 44.7214, 44.7214)):
    """Small SE2 analogue (boustrophedon grid walk with cross-row closures) used by tests."""
    N = int(n_poses)
    i = np.arange(N)
    row, col = i // side, i % side
    col = np.where(row % 2 == 0, col, side - 1 - col)
    x, y = col.astype(np.float64), row.astype(np.float64)
    th = np.where(row % 2 == 0, 0.0, np.pi - 1e-3) + 0.01 * np.sin(i)
    poses = np.stack([x, y, th], axis=-1)
    a = np.concatenate([np.arange(N - 1), np.arange(N - 2 * side + 1)[::3]]).astype(np.int32)
    b = a.copy()
    b[:N - 1] += 1
    nb = a[N - 1:]
    b[N - 1:] = (nb // side + 2) * side - 1 - (nb % side) - (side - 1) + (side - 1) - 0
    b = np.minimum(b, N - 1).astype(np.int32)
    keep = a != b
    a, b = a[keep], b[keep]
    E = len(a)
    c, s = np.cos(poses[a, 2]), np.sin(poses[a, 2])
    dx, dy = poses[b, 0] - poses[a, 0], poses[b, 1] - poses[a, 1]
    z = np.stack([c * dx + s * dy, -s * dx + c * dy, poses[b, 2] - poses[a, 2]], axis=-1)
    z += normal_stream(3 * E, seed).reshape(E, 3) / np.sqrt(np.asarray(info_diag))
    z[:, 2] = (z[:, 2] + np.pi) % (2 * np.pi) - np.pi
    info = np.zeros((E, 6))
    info[:, 0], info[:, 3], info[:, 5] = info_diag
    return {"pose_dim": 3, "ids": np.arange(N, dtype=np.int32), "poses": poses,
            "edge_ij": np.stack([a, b], axis=-1), "edge_data": np.concatenate([z, info], axis=-1)}
