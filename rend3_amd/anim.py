"""rend3-anim's API over the GPU pose path (rend3-anim/src/lib.rs): AnimationData::from_gltf_scene and
pose_animation_frame.  The per-skin work of pose_animation_frame -- sampling the channels, composing local matrices,
walking the joint hierarchy, multiplying by the inverse bind matrices -- runs on the GPU for every skeleton instance
(csrc/anim.hip, r3n_animation_write / r3n_pose_skeletons); this module flattens the loaded scene into the tables that
kernel reads and keeps the node-transform half (a handful of set_object_transform calls) on the host, where the
reference has it.

Host math is f32 with glam 0.25's formulas (Mat4::to_scale_rotation_translation for the bind components, Mat4::
from_scale_rotation_translation, Vec3 lerp / quaternion nlerp for the node transforms)."""
import numpy as np

f32 = np.float32
ONE, ZERO = f32(1.0), f32(0.0)


# ---------------------------------------------------------------------------------------------- glam restated (host side)
def _determinant(m):
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


def _quat_from_axes(xa, ya, za):
    m00, m01, m02 = xa
    m10, m11, m12 = ya
    m20, m21, m22 = za
    h = f32(0.5)
    if m22 <= ZERO:
        dif10, omm22 = f32(m11 - m00), f32(ONE - m22)
        if dif10 <= ZERO:
            f = f32(omm22 - dif10)
            i = f32(h / f32(np.sqrt(f)))
            return [f32(f * i), f32(f32(m01 + m10) * i), f32(f32(m02 + m20) * i), f32(f32(m12 - m21) * i)]
        f = f32(omm22 + dif10)
        i = f32(h / f32(np.sqrt(f)))
        return [f32(f32(m01 + m10) * i), f32(f * i), f32(f32(m12 + m21) * i), f32(f32(m20 - m02) * i)]
    sum10, opm22 = f32(m11 + m00), f32(ONE + m22)
    if sum10 <= ZERO:
        f = f32(opm22 - sum10)
        i = f32(h / f32(np.sqrt(f)))
        return [f32(f32(m02 + m20) * i), f32(f32(m12 + m21) * i), f32(f * i), f32(f32(m01 - m10) * i)]
    f = f32(opm22 + sum10)
    i = f32(h / f32(np.sqrt(f)))
    return [f32(f32(m12 - m21) * i), f32(f32(m20 - m02) * i), f32(f32(m01 - m10) * i), f32(f * i)]


def to_scale_rotation_translation(m):
    m = np.asarray(m, dtype=f32)
    det = _determinant(m)
    sign = f32(np.copysign(ONE, det)) if det == det else det

    def length3(v):
        return f32(np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))))

    scale = [f32(length3(m[0:3]) * sign), length3(m[4:7]), length3(m[8:11])]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = [f32(ONE / s) for s in scale]
    axes = [[f32(m[4 * c + k] * inv[c]) for k in range(3)] for c in range(3)]
    return np.array(scale, dtype=f32), np.array(_quat_from_axes(*axes), dtype=f32), m[12:15].copy()


def mat4_from_srt(s, q, t):
    x, y, z, w = (f32(v) for v in q)
    x2, y2, z2 = f32(x + x), f32(y + y), f32(z + z)
    xx, xy, xz, yy, yz, zz = f32(x * x2), f32(x * y2), f32(x * z2), f32(y * y2), f32(y * z2), f32(z * z2)
    wx, wy, wz = f32(w * x2), f32(w * y2), f32(w * z2)
    cols = ([f32(ONE - f32(yy + zz)), f32(xy + wz), f32(xz - wy), ZERO], [f32(xy - wz), f32(ONE - f32(xx + zz)), f32(yz + wx), ZERO],
            [f32(xz + wy), f32(yz - wx), f32(ONE - f32(xx + yy)), ZERO])
    m = np.zeros(16, dtype=f32)
    for c in range(3):
        for k in range(4):
            m[4 * c + k] = f32(cols[c][k] * f32(s[c]))
    m[12], m[13], m[14], m[15] = f32(t[0]), f32(t[1]), f32(t[2]), ONE
    return m


def _sample_index(times, t):
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


def _sample(channel, t, quat):
    times, values = channel
    p, n, x = _sample_index(np.asarray(times, dtype=f32), t)
    a, b = np.asarray(values[p], dtype=f32), np.asarray(values[n], dtype=f32)
    if not quat:
        return np.array([f32(a[k] + f32(f32(b[k] - a[k]) * x)) for k in range(3)], dtype=f32)
    dot = f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2])) + f32(a[3] * b[3])
    bias = ONE if dot >= ZERO else f32(-1.0)
    q = np.array([f32(a[k] + f32(f32(f32(b[k] * bias) - a[k]) * x)) for k in range(4)], dtype=f32)
    for _ in range(2):
        d = f32(f32(f32(q[0] * q[0]) + f32(q[1] * q[1])) + f32(q[2] * q[2])) + f32(q[3] * q[3])
        with np.errstate(invalid="ignore", divide="ignore"):
            r = ONE / f32(np.sqrt(f32(d)))
        q = np.array([f32(q[k] * r) for k in range(4)], dtype=f32)
    return q


# ---------------------------------------------------------------------------------------------- rend3-anim API
class AnimationData:
    """AnimationData::from_gltf_scene (rend3-anim/src/lib.rs:77-145) as flat tables: one rig per skin (joints in the
    skin's order, parent joint / hierarchy depth from the instance's node parents), one clip per (animation, skin) --
    pose_animation_frame poses EVERY skin of the instance with the chosen animation (:213), joints the animation does not
    touch at identity -- and the skeleton handles each skin deforms.  One AnimationData per scene instance, like the
    reference's."""

    def __init__(self, renderer, scene_animations, instance):
        self.instance = instance
        self.animations = scene_animations
        nodes = instance["nodes"]
        skins = instance["skins"]
        rigs = np.zeros(len(skins), dtype=[("first", np.uint32), ("n", np.uint32), ("depth", np.uint32), ("pad", np.uint32)])
        joint_dt = np.dtype([("parent", np.int32), ("depth", np.uint32), ("pad", np.uint32, 2), ("ibm", np.float32, 16)])
        track_dt = np.dtype([("animated", np.uint32), ("kf", np.uint32, 3), ("kc", np.uint32, 3), ("vf", np.uint32, 3),
                             ("bt", np.float32, 3), ("br", np.float32, 4), ("bs", np.float32, 3)])
        joints = []
        self.skin_skeletons = []
        for si, sk in enumerate(skins):
            node_to_joint = {n: j for j, n in enumerate(sk["joints"])}
            jrec = np.zeros(len(sk["joints"]), dtype=joint_dt)
            jrec["ibm"] = np.asarray(sk["inverse_bind_matrices"], dtype=f32).reshape(-1, 16)
            depth = {}
            for n in instance["topological_order"]:  # parents first
                if n not in node_to_joint:
                    continue
                j = node_to_joint[n]
                parent = nodes[n]["parent"]
                if parent is None:
                    jrec[j]["parent"], depth[j] = -1, 0
                elif parent not in node_to_joint:
                    jrec[j]["parent"], depth[j] = -2, 0
                else:
                    jrec[j]["parent"] = node_to_joint[parent]
                    depth[j] = depth[node_to_joint[parent]] + 1
                jrec[j]["depth"] = depth[j]
            rigs[si] = (sum(len(j) for j in joints), len(jrec), max(depth.values()) if depth else 0, 0)
            joints.append(jrec)
            self.skin_skeletons.append([h for nd in nodes if nd["skin"] == si for h in nd["skeletons"]])
        clips = np.zeros(len(scene_animations) * len(skins), dtype=[("rig", np.uint32), ("track", np.uint32), ("dur", np.float32), ("pad", np.uint32)])
        tracks, times, values = [], [], []
        n_times = n_values = 0
        for ai, anim in enumerate(scene_animations):
            for si, sk in enumerate(skins):
                trec = np.zeros(len(sk["joints"]), dtype=track_dt)
                for j, n in enumerate(sk["joints"]):
                    ch = anim["channels"].get(n)
                    if ch is None:
                        continue
                    trec[j]["animated"] = 1
                    bs, br, bt = to_scale_rotation_translation(nodes[n]["local_transform"])
                    trec[j]["bt"], trec[j]["br"], trec[j]["bs"] = bt, br, bs
                    for k, path in enumerate(("translation", "rotation", "scale")):
                        if path in ch:
                            t, v = ch[path]
                            t = np.ascontiguousarray(t, dtype=f32).reshape(-1)
                            v = np.ascontiguousarray(v, dtype=f32).reshape(-1)
                            trec[j]["kf"][k], trec[j]["kc"][k], trec[j]["vf"][k] = n_times, len(t), n_values
                            times.append(t)
                            values.append(v)
                            n_times += len(t)
                            n_values += len(v)
                clips[ai * len(skins) + si] = (si, sum(len(t) for t in tracks), anim["duration"], 0)
                tracks.append(trec)
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)  # noqa: E731
        self.n_skins = len(skins)
        self.clip_base = renderer.animation_add(rigs, cat(joints, joint_dt), clips, cat(tracks, track_dt), cat(times, f32), cat(values, f32))

    @classmethod
    def from_gltf_scene(cls, renderer, scene_animations, instance):
        return cls(renderer, scene_animations, instance)


def pose_animation_frame(renderer, instance, animation_data, animation_index, time):
    """pose_animation_frame (rend3-anim/src/lib.rs:181-263).  Node half on the host: every animated node's objects get
    the node's LOCAL matrix from the sampled scale / rotation / translation, z scale negated for a left-handed renderer
    (:191-211).  Skin half on the GPU: one pose request per skeleton of every skin."""
    anim = animation_data.animations[animation_index]
    t = f32(time)
    if t < ZERO:
        t = ZERO
    if t > anim["duration"]:
        t = f32(anim["duration"])
    nodes = instance["nodes"]
    for node_idx, ch in anim["channels"].items():
        if not nodes[node_idx]["objects"]:
            continue
        bs, br, bt = to_scale_rotation_translation(nodes[node_idx]["local_transform"])
        tr = _sample(ch["translation"], t, False) if "translation" in ch else bt
        ro = _sample(ch["rotation"], t, True) if "rotation" in ch else br
        sc = _sample(ch["scale"], t, False) if "scale" in ch else bs
        if renderer.handedness == 0:
            sc = np.array([sc[0], sc[1], f32(-sc[2])], dtype=f32)
        m = mat4_from_srt(sc, ro, tr)
        for h in nodes[node_idx]["objects"]:
            renderer.set_object_transform(h, m)
    requests = []
    for si in range(animation_data.n_skins):
        clip = animation_data.clip_base + animation_index * animation_data.n_skins + si
        requests += [(clip, t, sk) for sk in animation_data.skin_skeletons[si]]
    renderer.pose_skeletons(requests)
