"""oracle/anim.py -- CPU restatement of rend3-anim (TEST INFRASTRUCTURE, like the rest of oracle/).

Follows rend3-anim/src/lib.rs: sample_at_time (:163-175), the Lerp impls (:148-160: Vec3 lerp, quaternion nlerp +
normalize), pose_animation_frame (:181-263: node transforms from the sampled / bind TRS, joint local matrices, global
transforms in topological order, joint matrices = global * inverse bind through
Renderer::set_skeleton_joint_transforms, rend3/src/renderer/mod.rs:314-324).

Third-party math: glam 0.25 (not in /root/reference): Vec3::lerp = a + (b - a) * s; Quat::lerp = normalize(a + (b * bias
- a) * s) with bias = +-1 by the sign of the dot product; normalize = v * (1 / length); Mat4::from_scale_rotation_
translation via quat_to_axes (x2 = x + x ...); Mat4::to_scale_rotation_translation (determinant sign on the x scale,
Quat::from_rotation_axes); Mat4 * Mat4 column by column, ((x*X + y*Y) + z*Z) + w*W.  Scalar (non-SIMD) association of the
f32 sums.  PINNED on the reference's animation example screenshot (examples/src/animation: both scenes posed at t = 0,
tests/test_oracle_goldens.py::test_animation_example: silhouette identical in all 100 031 covered pixels) and on closed-form
cases (tests/test_anim.py).  The screenshot fixes the pose at t = 0 only; blending between keys rests on the closed-form
cases.

Every value is f32, one rounding per operation (numpy scalars)."""
import numpy as np

f32 = np.float32
ONE, ZERO, TWO = f32(1.0), f32(0.0), f32(2.0)


def _v(a):
    return np.asarray(a, dtype=f32)


def sample_index(times, t):
    """(prev, next, factor) of sample_at_time: next = first key with time > t (else the last), prev = next - 1 saturating;
    factor = clamp((t - t_prev) / (t_next - t_prev), 0, 1).  When both keys coincide (a time before the first key, a
    single-key channel) the division is x / 0: -inf / +inf clamp to 0 / 1 and give the key's value; 0 / 0 (t exactly on a
    lone key) is NaN and stays NaN through f32::clamp, poisoning the sample exactly like the reference."""
    times = _v(times)
    nxt = len(times) - 1
    for i, tk in enumerate(times):
        if tk > t:
            nxt = i
            break
    prv = max(nxt - 1, 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        x = f32(f32(t) - times[prv]) / f32(times[nxt] - times[prv])
    if x < ZERO:
        x = ZERO
    if x > ONE:
        x = ONE
    return prv, nxt, f32(x)


def lerp_vec3(a, b, s):
    a, b = _v(a), _v(b)
    return np.array([f32(a[k] + f32(f32(b[k] - a[k]) * s)) for k in range(3)], dtype=f32)


def _normalize4(q):
    d = f32(f32(f32(q[0] * q[0]) + f32(q[1] * q[1])) + f32(q[2] * q[2])) + f32(q[3] * q[3])
    with np.errstate(invalid="ignore", divide="ignore"):
        r = ONE / f32(np.sqrt(f32(d)))
    return np.array([f32(q[k] * r) for k in range(4)], dtype=f32)


def nlerp_quat(a, b, s):
    """impl Lerp for Quat: self.lerp(other, t).normalize(), Quat::lerp itself ending in a normalize."""
    a, b = _v(a), _v(b)
    dot = f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2])) + f32(a[3] * b[3])
    bias = ONE if dot >= ZERO else f32(-1.0)
    q = np.array([f32(a[k] + f32(f32(f32(b[k] * bias) - a[k]) * s)) for k in range(4)], dtype=f32)
    return _normalize4(_normalize4(q))


def sample_vec3(times, values, t):
    p, n, x = sample_index(times, t)
    return lerp_vec3(values[p], values[n], x)


def sample_quat(times, values, t):
    p, n, x = sample_index(times, t)
    return nlerp_quat(values[p], values[n], x)


def mat4_from_srt(s, q, t):
    """Mat4::from_scale_rotation_translation; column-major 16 floats."""
    s, q, t = _v(s), _v(q), _v(t)
    x, y, z, w = q
    x2, y2, z2 = f32(x + x), f32(y + y), f32(z + z)
    xx, xy, xz = f32(x * x2), f32(x * y2), f32(x * z2)
    yy, yz, zz = f32(y * y2), f32(y * z2), f32(z * z2)
    wx, wy, wz = f32(w * x2), f32(w * y2), f32(w * z2)
    xa = [f32(ONE - f32(yy + zz)), f32(xy + wz), f32(xz - wy), ZERO]
    ya = [f32(xy - wz), f32(ONE - f32(xx + zz)), f32(yz + wx), ZERO]
    za = [f32(xz + wy), f32(yz - wx), f32(ONE - f32(xx + yy)), ZERO]
    m = np.zeros(16, dtype=f32)
    for k in range(4):
        m[k] = f32(xa[k] * s[0])
        m[4 + k] = f32(ya[k] * s[1])
        m[8 + k] = f32(za[k] * s[2])
    m[12], m[13], m[14], m[15] = t[0], t[1], t[2], ONE
    return m


def mat4_mul(a, b):
    """glam Mat4 * Mat4: column c of the result = a * b.col(c) = ((a.x * bx + a.y * by) + a.z * bz) + a.w * bw."""
    a, b = _v(a), _v(b)
    out = np.zeros(16, dtype=f32)
    for c in range(4):
        for r in range(4):
            acc = f32(a[r] * b[4 * c])
            acc = f32(acc + f32(a[4 + r] * b[4 * c + 1]))
            acc = f32(acc + f32(a[8 + r] * b[4 * c + 2]))
            acc = f32(acc + f32(a[12 + r] * b[4 * c + 3]))
            out[4 * c + r] = acc
    return out


def determinant(m):
    """glam Mat4::determinant (cofactor expansion along the first column of the transposed layout)."""
    m = _v(m)
    m00, m01, m02, m03 = m[0:4]
    m10, m11, m12, m13 = m[4:8]
    m20, m21, m22, m23 = m[8:12]
    m30, m31, m32, m33 = m[12:16]
    a2323 = f32(f32(m22 * m33) - f32(m23 * m32))
    a1323 = f32(f32(m21 * m33) - f32(m23 * m31))
    a1223 = f32(f32(m21 * m32) - f32(m22 * m31))
    a0323 = f32(f32(m20 * m33) - f32(m23 * m30))
    a0223 = f32(f32(m20 * m32) - f32(m22 * m30))
    a0123 = f32(f32(m20 * m31) - f32(m21 * m30))
    t0 = f32(m00 * f32(f32(f32(m11 * a2323) - f32(m12 * a1323)) + f32(m13 * a1223)))
    t1 = f32(m01 * f32(f32(f32(m10 * a2323) - f32(m12 * a0323)) + f32(m13 * a0223)))
    t2 = f32(m02 * f32(f32(f32(m10 * a1323) - f32(m11 * a0323)) + f32(m13 * a0123)))
    t3 = f32(m03 * f32(f32(f32(m10 * a1223) - f32(m11 * a0223)) + f32(m12 * a0123)))
    return f32(f32(f32(t0 - t1) + t2) - t3)


def quat_from_rotation_axes(xa, ya, za):
    """glam Quat::from_rotation_axes (Mike Day, "Converting a Rotation Matrix to a Quaternion")."""
    m00, m01, m02 = xa
    m10, m11, m12 = ya
    m20, m21, m22 = za
    half = f32(0.5)
    if m22 <= ZERO:
        dif10 = f32(m11 - m00)
        omm22 = f32(ONE - m22)
        if dif10 <= ZERO:
            four_xsq = f32(omm22 - dif10)
            inv4x = f32(half / f32(np.sqrt(four_xsq)))
            return np.array([f32(four_xsq * inv4x), f32(f32(m01 + m10) * inv4x), f32(f32(m02 + m20) * inv4x), f32(f32(m12 - m21) * inv4x)], dtype=f32)
        four_ysq = f32(omm22 + dif10)
        inv4y = f32(half / f32(np.sqrt(four_ysq)))
        return np.array([f32(f32(m01 + m10) * inv4y), f32(four_ysq * inv4y), f32(f32(m12 + m21) * inv4y), f32(f32(m20 - m02) * inv4y)], dtype=f32)
    sum10 = f32(m11 + m00)
    opm22 = f32(ONE + m22)
    if sum10 <= ZERO:
        four_zsq = f32(opm22 - sum10)
        inv4z = f32(half / f32(np.sqrt(four_zsq)))
        return np.array([f32(f32(m02 + m20) * inv4z), f32(f32(m12 + m21) * inv4z), f32(four_zsq * inv4z), f32(f32(m01 - m10) * inv4z)], dtype=f32)
    four_wsq = f32(opm22 + sum10)
    inv4w = f32(half / f32(np.sqrt(four_wsq)))
    return np.array([f32(f32(m12 - m21) * inv4w), f32(f32(m20 - m02) * inv4w), f32(f32(m01 - m10) * inv4w), f32(four_wsq * inv4w)], dtype=f32)


def to_scale_rotation_translation(m):
    """Mat4::to_scale_rotation_translation -> (scale, rotation, translation)."""
    m = _v(m)
    det = determinant(m)
    sign = f32(np.copysign(ONE, det)) if det == det else det  # f32::signum: NaN stays NaN

    def length3(v):
        return f32(np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))))

    scale = np.array([f32(length3(m[0:3]) * sign), length3(m[4:7]), length3(m[8:11])], dtype=f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.array([f32(ONE / scale[k]) for k in range(3)], dtype=f32)
    axes = [np.array([f32(m[4 * c + k] * inv[c]) for k in range(3)], dtype=f32) for c in range(3)]
    rot = quat_from_rotation_axes(*axes)
    return scale, rot, m[12:15].copy()


IDENTITY = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], dtype=f32)


def sampled_trs(channels, bind_local, t):
    """(scale, rotation, translation) of one animated node at time t: the channel's sample, or the bind component of the
    node's local transform where the animation has no channel for that property (lib.rs:193-200, 229-236)."""
    bs, br, bt = to_scale_rotation_translation(bind_local)
    tr = sample_vec3(*channels["translation"], t) if channels.get("translation") else bt
    ro = sample_quat(*channels["rotation"], t) if channels.get("rotation") else br
    sc = sample_vec3(*channels["scale"], t) if channels.get("scale") else bs
    return sc, ro, tr


def pose_skin(animation, skin, nodes, topological_order, t):
    """Joint matrices of one skin at (clamped) time t.  animation: {"channels": {node: {...}}, "duration"}; skin:
    {"joints": [node indices], "inverse_bind_matrices": (n, 16)}; nodes: [{"local_transform", "parent"}].
    Joints the animation does not touch keep an IDENTITY local matrix (lib.rs:220: vec![Mat4::IDENTITY; n])."""
    joints = list(skin["joints"])
    node_to_joint = {n: j for j, n in enumerate(joints)}
    local = [IDENTITY.copy() for _ in joints]
    for node, ch in animation["channels"].items():
        if node not in node_to_joint:
            continue  # the reference indexes the map directly: an animated node outside the skin would panic there
        sc, ro, tr = sampled_trs(ch, nodes[node]["local_transform"], t)
        local[node_to_joint[node]] = mat4_from_srt(sc, ro, tr)
    glob = [IDENTITY.copy() for _ in joints]
    for node in [n for n in topological_order if n in node_to_joint]:
        j = node_to_joint[node]
        parent = nodes[node].get("parent")
        if parent is not None:
            pj = node_to_joint.get(parent)
            glob[j] = mat4_mul(glob[pj] if pj is not None else IDENTITY, local[j])
        else:
            glob[j] = local[j]
    ibm = np.asarray(skin["inverse_bind_matrices"], dtype=f32).reshape(len(joints), 16)
    return np.stack([mat4_mul(glob[j], ibm[j]) for j in range(len(joints))])


def clamp_time(animation, t):
    t = f32(t)
    d = f32(animation["duration"])
    if t < ZERO:
        t = ZERO
    if t > d:
        t = d
    return t


def node_object_matrix(channels, bind_local, t, left_handed):
    """The transform pose_animation_frame gives the objects of an animated node (lib.rs:191-211): the node's LOCAL matrix
    from the sampled TRS (not the global one), z scale negated for a left-handed renderer."""
    sc, ro, tr = sampled_trs(channels, bind_local, t)
    if left_handed:
        sc = np.array([sc[0], sc[1], f32(-sc[2])], dtype=f32)
    return mat4_from_srt(sc, ro, tr)


def pose_animation_frame(renderer, instance, animations, animation_index, time):
    """pose_animation_frame (rend3-anim/src/lib.rs:181-263) against an OracleRenderer: node transforms, then every skin's
    joint matrices through set_skeleton_joint_matrices (= set_skeleton_joint_transforms, renderer/mod.rs:314-324)."""
    anim = animations[animation_index]
    t = clamp_time(anim, time)
    nodes = instance["nodes"]
    for node_idx, ch in anim["channels"].items():
        if not nodes[node_idx]["objects"]:
            continue
        m = node_object_matrix(ch, nodes[node_idx]["local_transform"], t, renderer.handedness == 0)
        for h in nodes[node_idx]["objects"]:
            renderer.set_object_transform(h, m)
    for si, skin in enumerate(instance["skins"]):
        mats = pose_skin(anim, skin, nodes, instance["topological_order"], t)
        for nd in nodes:
            if nd["skin"] == si:
                for sk in nd["skeletons"]:
                    renderer.set_skeleton_joint_matrices(sk, mats)
