"""Exact float64 ray tracing of photon's single thick-lens camera, for the tests that hold the library's static bounds
(live lens samples, culled sources) against the geometry itself.  An independent restatement in numpy of what the kernels
do per ray (generate_lightfield_angular_data .cu:123-141, the 'l' element .cu:507-864, the sensor plane .cu:1383-1452,
1735-1815); test infrastructure only."""
import numpy as np


def lens_samples(photon_or_oracle, call):
    """(x_lens, y_lens) of every lens sample of the call: ratio * pitch * r1 * (cos, sin)(2 pi r2), .cu:123-124."""
    r1, r2 = photon_or_oracle.rand_table(int(call.lightray_number_per_particle))
    r = float(call.ray_cone_pitch_ratio) * float(call.lens_pitch) * r1.astype(np.float64)
    ang = 2 * np.pi * r2.astype(np.float64)
    return r * np.cos(ang), r * np.sin(ang)


def _sphere(pos, d, cz, R):
    o = pos - np.array([0.0, 0.0, cz])
    a = (d * d).sum(-1)
    b = 2 * (d * o).sum(-1)
    c = (o * o).sum(-1) - R * R
    with np.errstate(invalid="ignore"):
        root = np.sqrt(b * b - 4 * a * c)
    t1, t2 = (-b + root) / (2 * a), (-b - root) / (2 * a)
    t = np.minimum(t1, t2) if R > 0 else np.maximum(t1, t2)             # .cu:298-336
    return pos + d * t[..., None]


def trace_thick_lens(sources, px, py, z_aim, zc, t, R1, R2, n, pitch, z_sensor, margin=0.0):
    """sources [ns, 3], lens samples px/py [np] on the plane z = z_aim; one 'l' element centred on the z axis at zc with normal
    +z.  Returns (hits [ns, np, 2], alive [ns, np]): where each ray meets the plane z = z_sensor and whether it passed both
    aperture tests (axis distance <= pitch / 2 + margin) and both refractions."""
    S = np.asarray(sources, np.float64)
    ns, npnt = S.shape[0], len(px)
    d = np.stack([px[None, :] - S[:, 0:1], py[None, :] - S[:, 1:2], np.broadcast_to(z_aim - S[:, 2:3], (ns, npnt))], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    pos = np.broadcast_to(S[:, None, :], d.shape)
    with np.errstate(invalid="ignore", divide="ignore"):
        c1 = zc + t / 2 - R1                                            # .cu:557
        H1 = _sphere(pos, d, c1, R1)
        alive = np.hypot(H1[..., 0], H1[..., 1]) <= pitch / 2 + margin
        N = H1 - np.array([0.0, 0.0, c1])
        N /= np.linalg.norm(N, axis=-1, keepdims=True)
        eta = 1.0 / n
        cosi = -(d * N).sum(-1)
        rad = 1 - eta * eta * (1 - cosi * cosi)
        alive &= rad >= 0
        v = d * eta + (eta * cosi - np.sqrt(np.abs(rad)))[..., None] * N
        v /= np.linalg.norm(v, axis=-1, keepdims=True)
        c2 = zc - t / 2 - R2                                            # .cu:704
        H2 = _sphere(H1, v, c2, R2)
        alive &= np.hypot(H2[..., 0], H2[..., 1]) <= pitch / 2 + margin
        N2 = -(H2 - np.array([0.0, 0.0, c2]))
        N2 /= np.linalg.norm(N2, axis=-1, keepdims=True)
        cosi = -(v * N2).sum(-1)
        rad = 1 - n * n * (1 - cosi * cosi)
        alive &= rad >= 0
        w = n * v + (n * cosi - np.sqrt(np.abs(rad)))[..., None] * N2
        w /= np.linalg.norm(w, axis=-1, keepdims=True)
        tt = (z_sensor - H2[..., 2]) / w[..., 2]
        hits = H2[..., :2] + w[..., :2] * tt[..., None]
    alive &= np.isfinite(hits).all(-1)
    return hits, alive


def call_lens(call):
    """(zc, t, R1, R2, n, pitch) of the call's element 0."""
    e = call.elements[0]
    g = e["element_geometry"]
    return (float(call.element_center[0][2]), float(g["vertex_distance"]), float(g["front_surface_radius"]),
            float(g["back_surface_radius"]), float(e["element_properties"]["refractive_index"]), float(g["pitch"]))


def trace_thin_lens(sources, px, py, z_aim, z_plane, focal, pitch, z_sensor, margin=0.0):
    """The 't' element (.cu:416-503): the ray meets the plane z = z_plane (the element's centre lies on it, on the axis) at H,
    dies beyond pitch / 2 + margin from the axis, leaves along u - (H - C) / f.  Returns (hits [ns, np, 2], alive [ns, np])."""
    S = np.asarray(sources, np.float64)
    ns, npnt = S.shape[0], len(px)
    d = np.stack([px[None, :] - S[:, 0:1], py[None, :] - S[:, 1:2], np.broadcast_to(z_aim - S[:, 2:3], (ns, npnt))], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        t = (z_plane - S[:, None, 2]) / d[..., 2]
        H = S[:, None, :] + d * t[..., None]
        alive = np.hypot(H[..., 0], H[..., 1]) <= pitch / 2 + margin
        w = d - (H - np.array([0.0, 0.0, z_plane])) / focal
        w /= np.linalg.norm(w, axis=-1, keepdims=True)
        tt = (z_sensor - H[..., 2]) / w[..., 2]
        hits = H[..., :2] + w[..., :2] * tt[..., None]
    alive &= np.isfinite(hits).all(-1)
    return hits, alive
