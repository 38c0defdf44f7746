"""Row N4, rend3-anim: the oracle's restatement (oracle/anim.py) on closed-form cases, the glTF animation loader, the
flattened rig / clip tables of rend3_amd/anim.py, and -- on the GPU -- the pose kernel (csrc/anim.hip) against the oracle:
joint matrices bit for bit, then whole animated frames."""
import math
import os

import numpy as np
import pytest

import scenes
from oracle import anim as oa
from oracle import host as oh
from oracle.world import OracleRenderer, material_record as omk
from rend3_amd import anim as pa
from rend3_amd.gltf import Gltf, instance_scene, load_animations

f32 = np.float32


def qz(a):
    return np.array([0, 0, math.sin(a / 2), math.cos(a / 2)], dtype=f32)


def test_sample_at_time_semantics():
    times = np.array([0.5, 1.0, 2.0], dtype=f32)
    assert oa.sample_index(times, f32(1.0))[:2] == (1, 2) and oa.sample_index(times, f32(1.0))[2] == 0.0   # a key's own time starts its segment
    p, n, x = oa.sample_index(times, f32(1.5))
    assert (p, n) == (1, 2) and x == f32(0.5)
    p, n, x = oa.sample_index(times, f32(3.0))      # past the end: the last segment, factor clamped to 1
    assert (p, n) == (1, 2) and x == f32(1.0)
    p, n, x = oa.sample_index(times, f32(0.25))     # before the first key: both keys coincide, -0.25 / 0 = -inf -> clamped to 0
    assert (p, n) == (0, 0) and x == 0.0
    assert np.array_equal(oa.sample_vec3(times, np.arange(9, dtype=f32).reshape(3, 3), f32(0.0)), np.array([0, 1, 2], f32))
    one = np.array([1.0], dtype=f32)                # single-key channel: value before / after the key, 0 / 0 = NaN exactly on it
    assert oa.sample_index(one, f32(0.5))[2] == 0.0 and oa.sample_index(one, f32(2.0))[2] == 1.0
    assert np.isnan(oa.sample_index(one, f32(1.0))[2]) and np.isnan(oa.sample_vec3(one, np.ones((1, 3), f32), f32(1.0))).all()


def test_lerp_nlerp_and_matrices():
    assert np.array_equal(oa.lerp_vec3([0, 2, 4], [2, 2, 0], f32(0.25)), np.array([0.5, 2, 3], f32))
    # nlerp takes the short way round (bias -1 when the dot product is negative) and returns a unit quaternion
    q = oa.nlerp_quat(qz(0.2), -qz(0.6), f32(0.5))
    assert abs(float(np.dot(q, q)) - 1.0) < 1e-6 and np.allclose(q, qz(0.4), atol=1e-3)
    # 90 degrees about z, scale (2, 3, 4), translation (5, 6, 7): columns = rotated axes times the scale
    m = oa.mat4_from_srt([2, 3, 4], qz(math.pi / 2), [5, 6, 7]).reshape(4, 4)
    assert np.allclose(m[0], [0, 2, 0, 0], atol=1e-6) and np.allclose(m[1], [-3, 0, 0, 0], atol=1e-6)
    assert np.allclose(m[2], [0, 0, 4, 0]) and np.allclose(m[3], [5, 6, 7, 1])
    # round trip, including a mirrored basis (negative determinant -> negative x scale)
    for s in ([2, 3, 4], [-2, 3, 4]):
        m = oa.mat4_from_srt(s, qz(0.7), [1, 2, 3])
        sc, ro, tr = oa.to_scale_rotation_translation(m)
        assert np.allclose(sc, s, atol=1e-5) and np.allclose(tr, [1, 2, 3])
        assert np.allclose(oa.mat4_from_srt(sc, ro, tr), m, atol=1e-5)
    a, b = oa.mat4_from_srt([1, 1, 1], qz(0.3), [1, 0, 0]), oa.mat4_from_srt([1, 2, 1], qz(-0.2), [0, 1, 0])
    assert np.allclose(oa.mat4_mul(a, b).reshape(4, 4).T, a.reshape(4, 4).T @ b.reshape(4, 4).T, atol=1e-6)
    # the product's host copy of the decomposition is the same arithmetic
    for m in (a, b, oa.mat4_from_srt([-1.5, 0.5, 2], qz(2.5), [3, 2, 1])):
        for x, y in zip(oa.to_scale_rotation_translation(m), pa.to_scale_rotation_translation(m)):
            assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))


def chain_skin():
    ident = oa.IDENTITY
    nodes = [dict(parent=None, local_transform=ident), dict(parent=0, local_transform=oa.mat4_from_srt([1, 1, 1], qz(0), [1, 0, 0])),
             dict(parent=1, local_transform=oa.mat4_from_srt([1, 1, 1], qz(0), [1, 0, 0])), dict(parent=9, local_transform=ident)]
    skin = dict(joints=[0, 1, 2, 3], inverse_bind_matrices=np.tile(ident, (4, 1)))
    return nodes, skin


def test_pose_skin_hierarchy_and_quirks():
    nodes, skin = chain_skin()
    nodes.append(dict(parent=None, local_transform=oa.IDENTITY))
    tr = (np.array([0, 1], f32), np.array([[0, 0, 0], [0, 2, 0]], f32))
    anim = dict(channels={0: dict(translation=tr), 2: dict(rotation=(np.array([0, 1], f32), np.stack([qz(0), qz(math.pi)])))}, duration=f32(1.0))
    nodes[3]["parent"] = 4  # parent node exists but is not a joint of the skin
    mats = oa.pose_skin(anim, skin, nodes, [4, 0, 1, 2, 3], f32(0.5))
    # joint 0 moved to y = 1; joint 1 is NOT animated: its local matrix is IDENTITY (not its bind translation), so it sits on
    # joint 0; joint 2 rotates in place; joint 3 (parent outside the skin) = IDENTITY * IDENTITY
    assert np.allclose(mats[0].reshape(4, 4)[3], [0, 1, 0, 1]) and np.allclose(mats[1].reshape(4, 4)[3], [0, 1, 0, 1])
    assert np.allclose(mats[2].reshape(4, 4)[3], [1, 1, 0, 1]) and np.allclose(mats[2].reshape(4, 4)[0][:2], [0, 1], atol=1e-6)
    assert np.array_equal(mats[3], oa.IDENTITY)
    assert oa.clamp_time(anim, 5.0) == f32(1.0) and oa.clamp_time(anim, -1.0) == 0.0


class FakeRenderer:
    handedness = 0

    def animation_add(self, *tables):
        self.tables = tables
        return 0


def load_animated(r, hm, mk, tmp_path):
    g = Gltf(scenes.write_animated_gltf(os.path.join(str(tmp_path), "animated.gltf")))
    inst = instance_scene(g, r, hm, mk)
    anims = load_animations(g)
    r.add_directional_light(color=(1, 1, 1), intensity=4.0, direction=(-0.3, -0.6, 1.0), distance=12.0, resolution=256)
    r.set_camera_data(hm.look_at_lh((0.5, 1.5, -8.0), (0, 0.5, 0), (0, 1, 0)), ("perspective", 50.0, 0.1))
    return g, inst, anims


def test_gltf_animation_loading_and_tables(tmp_path):
    o = OracleRenderer(oh.LEFT, f32(320) / f32(192))
    g, inst, anims = load_animated(o, oh, omk, tmp_path)
    assert len(anims) == 1 and anims[0]["duration"] == f32(2.0) and sorted(anims[0]["channels"]) == [1, 2, 3, 4, 7]
    assert set(anims[0]["channels"][3]) == {"rotation", "scale"} and len(anims[0]["channels"][3]["rotation"][0]) == 2
    assert inst["skins"][0]["joints"] == [1, 2, 3, 4, 5] and inst["nodes"][6]["skin"] == 0 and len(inst["nodes"][6]["skeletons"]) == 1
    assert inst["nodes"][7]["objects"] and inst["nodes"][2]["parent"] == 1 and inst["nodes"][1]["parent"] == 0
    fake = FakeRenderer()
    data = pa.AnimationData.from_gltf_scene(fake, anims, inst)
    rigs, joints, clips, tracks, times, values = fake.tables
    assert rigs.tolist() == [(0, 5, 3, 0)] and clips["rig"].tolist() == [0] and clips["dur"][0] == f32(2.0)
    # joint 0's parent (node 0) is not a joint -> -2; the chain 1 <- 2 <- 3; the side joint hangs off joint 1
    assert joints["parent"].tolist() == [-2, 0, 1, 2, 1] and joints["depth"].tolist() == [0, 1, 2, 3, 2]
    assert tracks["animated"].tolist() == [1, 1, 1, 1, 0]
    assert tracks["kc"].tolist() == [[4, 0, 0], [0, 4, 0], [0, 2, 4], [0, 4, 0], [0, 0, 0]]
    assert len(times) == 4 + 4 + 2 + 4 + 4 and len(values) == 12 + 16 + 8 + 12 + 16
    assert data.skin_skeletons == [inst["nodes"][6]["skeletons"]]
    # bind components of an animated joint: node 3 has translation (1, 0, 0), unit scale, identity rotation
    assert np.allclose(tracks["bt"][2], [1, 0, 0]) and np.allclose(tracks["br"][2], [0, 0, 0, 1]) and np.allclose(tracks["bs"][2], [1, 1, 1])


def test_oracle_animated_frames_move(tmp_path):
    o = OracleRenderer(oh.LEFT, f32(160) / f32(96))
    g, inst, anims = load_animated(o, oh, omk, tmp_path)
    frames = []
    for t in (0.4, 1.2):
        oa.pose_animation_frame(o, inst, anims, 0, t)
        frames.append(o.render(160, 96, ambient=(0.2, 0.2, 0.2, 1.0), clear_color=(0, 0, 0, 1)))
    assert ((frames[0]["vis"] & np.uint64(0xFFFFFFFF)) != 0).mean() > 0.02
    assert (frames[0]["vis"] != frames[1]["vis"]).mean() > 0.01  # the bar bends and the cube moves


# ------------------------------------------------------------------------------------------------ GPU
def random_rig_case(rng, n_joints, n_extra_nodes=2):
    """A random node forest with `n_joints` joints (some with parents outside the skin), a clip animating a random
    subset of them with random subsets of channels (some single-key, some starting late), as instance / animation dicts."""
    n_nodes = n_joints + n_extra_nodes
    order = list(range(n_nodes))
    nodes = []
    for i in order:
        parent = None if i == 0 or rng.random() < 0.1 else int(rng.integers(0, i))
        sc = rng.uniform(0.5, 1.5, 3) * (-1 if rng.random() < 0.1 else 1)
        q = rng.normal(size=4)
        q = (q / np.linalg.norm(q)).astype(f32)
        local = oa.mat4_from_srt(sc.astype(f32), q, rng.uniform(-2, 2, 3).astype(f32))
        nodes.append(dict(parent=parent, local_transform=local, objects=[], skin=None, skeletons=[]))
    joint_nodes = sorted(rng.choice(n_nodes, n_joints, replace=False).tolist())
    rng.shuffle(joint_nodes)
    ibm = np.stack([oa.mat4_from_srt([1, 1, 1], qz(rng.uniform(-1, 1)), rng.uniform(-1, 1, 3).astype(f32)) for _ in joint_nodes])
    skin = dict(joints=joint_nodes, inverse_bind_matrices=ibm)
    channels = {}
    for n in joint_nodes:
        if rng.random() < 0.25:
            continue
        ch = {}
        for path, width in (("translation", 3), ("rotation", 4), ("scale", 3)):
            if rng.random() < 0.35:
                continue
            k = 1 if rng.random() < 0.08 else int(rng.integers(2, 7))
            times = np.sort(rng.uniform(0.0 if rng.random() < 0.7 else 0.4, 3.0, k)).astype(f32)
            if k > 2 and rng.random() < 0.15:
                times[[0, k - 1]] = times[[k - 1, 0]]   # an unsorted channel: the reference's linear scan decides, not bisection
            if k > 3 and rng.random() < 0.15:
                times[1] = times[2]                      # duplicate key times
            vals = rng.normal(size=(k, width)).astype(f32)
            if path == "rotation":
                vals = (vals / np.linalg.norm(vals, axis=1, keepdims=True)).astype(f32)
            if path == "scale":
                vals = np.abs(vals) + f32(0.3)
            ch[path] = (times, vals)
        if ch:
            channels[n] = ch
    dur = max([float(t[0].max()) for c in channels.values() for t in c.values()] + [0.0])
    return dict(nodes=nodes, skins=[skin], topological_order=order), dict(channels=channels, duration=f32(dur))


@pytest.mark.gpu
def test_gpu_pose_kernel_matches_oracle():
    """csrc/anim.hip against oracle/anim.py: joint matrices of random rigs (1 ... 70 joints: more joints than lanes, deep
    and shallow hierarchies, parents outside the skin, mirrored bind transforms), clips with missing channels,
    single-key and late-starting channels (NaN samples), times below, inside and beyond the clip: bit-identical,
    NaNs included."""
    import rend3_amd as r3
    rng = np.random.default_rng(0xA11CE)
    for n_joints in (1, 2, 5, 23, 64, 70):
        inst, anim = random_rig_case(rng, n_joints)
        p = r3.Renderer()
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], f32)
        mesh = p.add_mesh(pos, np.array([0, 1, 2], np.uint32), joint_indices=np.zeros((3, 4), np.uint16),
                          joint_weights=np.tile(np.array([1, 0, 0, 0], f32), (3, 1)))
        times = [-0.5, 0.0, 0.3, 1.1, float(anim["duration"]), float(anim["duration"]) + 1.0]
        sks = p.add_skeletons_bulk(mesh, [np.tile(oh.identity(), (n_joints, 1))] * len(times))
        inst["nodes"][0]["skin"], inst["nodes"][0]["skeletons"] = 0, sks
        data = pa.AnimationData.from_gltf_scene(p, [anim], inst)
        assert data.skin_skeletons == [sks]
        p.pose_skeletons([(0, t, sk) for t, sk in zip(times, sks)])
        mat = p.add_material(r3.material_record(albedo=(1, 1, 1, 1)))
        p.add_object(None, mat, oh.identity(), skeleton=sks[0])
        p.set_camera_data(oh.look_at_lh((0, 0, -3), (0, 0, 0), (0, 1, 0)), ("perspective", 60.0, 0.1))
        p.render(32, 32, readback=False)
        got = p.readback_joint_matrices().reshape(len(times), n_joints, 16)
        for k, t in enumerate(times):
            want = oa.pose_skin(anim, inst["skins"][0], inst["nodes"], inst["topological_order"], oa.clamp_time(anim, t))
            assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), f"{n_joints} joints, t = {t}: {(got[k] != want).sum()} floats differ"
        p.close()


@pytest.mark.gpu
def test_gpu_animated_gltf_frames_match_oracle(tmp_path):
    """pose_animation_frame end to end on the animated glTF: node transforms on the host, joint matrices on the GPU,
    skinning, cull, raster, shade -- every frame bit-identical to the oracle posed by oracle/anim.py; the pose persists
    over a frame without a new pose call, and explicit joint matrices take over again afterwards."""
    import rend3_amd as r3
    from test_gpu_parity import compare_frames
    o, p = OracleRenderer(oh.LEFT, f32(320) / f32(192)), r3.Renderer(oh.LEFT, f32(320) / f32(192))
    _g, inst_o, anims = load_animated(o, oh, omk, tmp_path)
    _g, inst_p, anims_p = load_animated(p, r3.host, r3.material_record, tmp_path)
    data = pa.AnimationData.from_gltf_scene(p, anims_p, inst_p)
    kw = dict(ambient=(0.2, 0.2, 0.2, 1.0), clear_color=(0.02, 0.03, 0.05, 1.0))
    for f, t in enumerate((0.4, 0.9, 1.7, 2.5)):
        oa.pose_animation_frame(o, inst_o, anims, 0, t)
        pa.pose_animation_frame(p, inst_p, data, 0, t)
        compare_frames(o.render(320, 192, **kw), p.render(320, 192, **kw), f"animated frame {f} (t = {t})")
    compare_frames(o.render(320, 192, **kw), p.render(320, 192, **kw), "pose persists")
    ident = np.tile(oh.identity(), (5, 1))
    o.set_skeleton_joint_matrices(inst_o["skeletons"][0], ident)
    p.set_skeleton_joint_matrices(inst_p["skeletons"][0], ident)
    compare_frames(o.render(320, 192, **kw), p.render(320, 192, **kw), "explicit matrices after a pose")


@pytest.mark.gpu
def test_gpu_animation_example_matches_oracle():
    """The reference's animation example (34-joint character + animated cube, tests/golden/animation-*.glb) at three times:
    poses on the GPU, every frame bit-identical to the oracle at 640x360."""
    import rend3_amd as r3
    import test_oracle_goldens as G
    from test_gpu_parity import compare_frames
    o, p = OracleRenderer(oh.LEFT, f32(640) / f32(360)), r3.Renderer(oh.LEFT, f32(640) / f32(360))
    so = G.build_animation_example(o, oh, omk)
    sp = [(inst, anims, pa.AnimationData.from_gltf_scene(p, anims, inst)) for inst, anims in G.build_animation_example(p, r3.host, r3.material_record)]
    assert [d.clip_base for _i, _a, d in sp] == [0, 1]  # one AnimationData per scene instance, sharing the library's tables
    kw = dict(clear_color=(0.10, 0.05, 0.10, 1.0))
    for t in (0.0, 3.3, 7.9):
        for inst, anims in so:
            oa.pose_animation_frame(o, inst, anims, 0, t)
        for inst, _anims, data in sp:
            pa.pose_animation_frame(p, inst, data, 0, t)
        compare_frames(o.render(640, 360, **kw), p.render(640, 360, **kw), f"animation example t = {t}")
