"""Row S1 (GPU skinning, rend3-routine/shaders/src/skinning.wgsl + skinning.rs): oracle known-answer tests on CPU,
HIP-vs-oracle bit-exact parity and a config-5-shaped run (many instances, one launch) on the GPU."""
import math

import numpy as np
import pytest

import scenes
from oracle import host as oh
from oracle.world import OracleRenderer
from oracle.world import material_record as omk
from rend3_amd.scenes import Pcg32, skinned_cylinder

f32 = np.float32


def _rig(r, joints, mk, n_skeletons=1):
    pos, idx, nrm, tang, ji, jw = skinned_cylinder(joints)
    mesh = r.add_mesh(pos, idx, normals=nrm, tangents=tang, joint_indices=ji, joint_weights=jw)
    mat = r.add_material(mk(albedo=(0.8, 0.6, 0.4, 1.0), albedo_mode="value", roughness=0.5), 0)
    sks = [r.add_skeleton(mesh, np.tile(oh.identity(), (joints, 1))) for _ in range(n_skeletons)]
    return mesh, mat, sks, pos, nrm


def _pose(joints, seed, scale=True):
    rng = Pcg32(seed)
    mats = []
    for j in range(joints):
        m = oh.mat4_mul(oh.translation((rng.uniform(-0.2, 0.2), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2))),
                        scenes.random_rotation(rng, oh))
        if scale:
            m = oh.mat4_mul(m, oh.scale((rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0))))
        mats.append(m)
    return np.array(mats, dtype=f32)


def _skinned_runs(r, sk):
    s = r.skeletons[sk]
    n = 3 * r.meshes[s["mesh"]].vertex_count
    if hasattr(r, "mesh_words"):
        return [r.mesh_words[o // 4:o // 4 + n].copy() for o in s["out_off"]]
    return [r.readback_mesh_words(o, n) for o in s["out_off"]]


# ------------------------------------------------------------------ CPU: oracle known answers
def test_oracle_identity_pose_reproduces_bind_pose():
    o = OracleRenderer(oh.LEFT)
    _mesh, mat, (sk,), pos, nrm = _rig(o, 4, omk)
    o.add_object(None, mat, oh.identity(), skeleton=sk)
    o.set_camera_data(oh.look_at_lh((0, 1, -4), (0, 1, 0), (0, 1, 0)), ("perspective", 60.0, 0.1))
    o.render(64, 64)
    p, n, t = (a.view(f32).reshape(-1, 3) for a in _skinned_runs(o, sk))
    # weights of each vertex sum to exactly 1 in f32 for this rig, identity matrices => bind pose back
    assert np.allclose(p, pos, atol=1e-6) and np.allclose(n, nrm, atol=1e-6)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)


def test_oracle_translation_and_nonuniform_scale():
    o = OracleRenderer(oh.LEFT)
    _mesh, mat, (sk,), pos, nrm = _rig(o, 2, omk)
    o.add_object(None, mat, oh.identity(), skeleton=sk)
    o.set_camera_data(oh.look_at_lh((0, 1, -4), (0, 1, 0), (0, 1, 0)), ("perspective", 60.0, 0.1))
    m = oh.mat4_mul(oh.translation((1.0, 2.0, 3.0)), oh.scale((2.0, 1.0, 0.5)))
    o.set_skeleton_joint_matrices(sk, np.array([m, m], dtype=f32))
    o.render(64, 64)
    p, n, _t = (a.view(f32).reshape(-1, 3) for a in _skinned_runs(o, sk))
    assert np.allclose(p, pos * np.array([2.0, 1.0, 0.5], dtype=f32) + np.array([1, 2, 3], dtype=f32), atol=1e-5)
    # normals go through J3 * (inv_scale^2 o n) == inverse-transpose for a pure scale (skinning.wgsl:75-76)
    expect = nrm / np.array([2.0, 1.0, 0.5], dtype=f32)
    expect /= np.linalg.norm(expect, axis=1, keepdims=True)
    assert np.allclose(n, expect, atol=1e-5)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("joints", [2, 7, 34])
def test_gpu_skinning_bit_exact_and_rendered(joints):
    import torch
    assert torch.cuda.is_available()
    import rend3_amd as r3
    from test_gpu_parity import compare_frames
    o, p = OracleRenderer(oh.LEFT, f32(1.5)), r3.Renderer(oh.LEFT, f32(1.5))
    rigs = []
    for r, mk in ((o, omk), (p, r3.material_record)):
        _mesh, mat, sks, _pos, _nrm = _rig(r, joints, mk, n_skeletons=3)
        for i, sk in enumerate(sks):
            r.add_object(None, mat, oh.translation((-1.2 + 1.2 * i, 0.0, 0.0)), skeleton=sk)
        r.add_directional_light(color=(1, 1, 1), intensity=3.0, direction=(0.3, -1.0, 0.4), distance=10.0, resolution=256)
        r.set_camera_data(oh.look_at_lh((0, 1.2, -4), (0, 1, 0), (0, 1, 0)), ("perspective", 60.0, 0.1))
        rigs.append(sks)
    for f in range(3):
        for r, sks in zip((o, p), rigs):
            for i, sk in enumerate(sks):
                r.set_skeleton_joint_matrices(sk, _pose(joints, 100 * f + i, scale=(f != 1)))
        fo, fp = o.render(192, 128, ambient=(0.1, 0.1, 0.1, 1)), p.render(192, 128, ambient=(0.1, 0.1, 0.1, 1))
        for i in range(3):
            for a, b in zip(_skinned_runs(o, rigs[0][i]), _skinned_runs(p, rigs[1][i])):
                assert np.array_equal(a, b), f"skinned attribute run differs (frame {f}, skeleton {i})"
        compare_frames(fo, fp, f"skinned frame {f}")
        assert fo["pass"].sum() > 0
    p.close()


@pytest.mark.gpu
def test_gpu_config5_shape_many_instances():
    """BASELINE.json configs[4] shape, reduced instance count for test time: 5 000 skeletons x 192 vertices x 2 joints
    skinned by ONE launch; every instance posed differently; spot-check 16 instances bit-exact against the oracle."""
    import torch
    assert torch.cuda.is_available()
    import rend3_amd as r3
    n = 5000
    p = r3.Renderer(oh.LEFT, f32(16 / 9))
    pos, idx, nrm, tang, ji, jw = skinned_cylinder(2)
    assert len(pos) == 192 and len(np.unique(ji)) == 2  # the generator this test relies on: 192 vertices per instance, 2 joints
    mesh = p.add_mesh(pos, idx, normals=nrm, tangents=tang, joint_indices=ji, joint_weights=jw)
    mat = p.add_material(r3.material_record(albedo=(0.7, 0.7, 0.7, 1), albedo_mode="value", roughness=0.6), 0)
    poses = [_pose(2, 7 + i) for i in range(n)]
    sks = p.add_skeletons_bulk(mesh, poses)
    rng = np.random.Generator(np.random.PCG64(5))
    xf = np.tile(oh.identity(), (n, 1))
    xf[:, 12] = rng.uniform(-60, 60, n); xf[:, 14] = rng.uniform(5, 120, n)
    for i, sk in enumerate(sks):
        p.add_object(None, mat, xf[i], skeleton=sk)
    p.set_camera_data(oh.look_at_lh((0, 10, -10), (0, 0, 40), (0, 1, 0)), ("perspective", 60.0, 0.1))
    out = p.render(640, 360)
    assert out["pass"].sum() > 0
    o = OracleRenderer(oh.LEFT)
    omesh = o.add_mesh(pos, idx, normals=nrm, tangents=tang, joint_indices=ji, joint_weights=jw)
    for i in (0, 1, 2, 3, 17, 99, 1000, 1001, 2500, 3333, 4000, 4095, 4096, 4997, 4998, 4999):
        osk = o.add_skeleton(omesh, poses[i])
        sk_in, sk_m = o.skinning_buffers()
        o.lib.r3o_skinning(o.lib.ptr(o.mesh_words), o.lib.ptr(sk_in), len(sk_in), o.lib.ptr(sk_m))
        for a, b in zip(_skinned_runs(o, osk), _skinned_runs(p, sks[i])):
            assert np.array_equal(a, b), i
    p.close()


@pytest.mark.gpu
def test_gpu_config5_full_size():
    """BASELINE.json configs[4] at FULL size on the asset it names: the reference's examples/src/skinning/RiggedSimple.glb
    (160 vertices, 2 joints; tests/golden/skinning-RiggedSimple.glb, read by the product's GLB reader) x 50 000 skeleton
    instances, every instance with its own pair of joint matrices (rotation + translation + non-uniform scale), skinned by
    ONE launch (skinning.rs:142-199 issues 50 000 dispatches).  EVERY skinned position / normal run of EVERY instance --
    8 000 000 vertices -- is compared bit for bit with the oracle's restatement of skinning.wgsl:37-94 over the same
    GpuSkinningInput records (skinning.rs:23-46) and matrices."""
    import os
    import torch
    assert torch.cuda.is_available()
    import rend3_amd as r3
    from rend3_amd.gltf import Gltf
    n = 50_000
    g = Gltf(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skinning-RiggedSimple.glb"))
    prim = g.primitive(0, 0)
    assert len(prim["positions"]) == 160 and int(prim["joints"].max()) == 1
    idx = prim["indices"].reshape(-1, 3)[:, ::-1].reshape(-1)  # left-handed renderer: load_meshes flips the winding (rend3-gltf/src/lib.rs:628-634)
    kw = dict(normals=prim.get("normals"), tangents=prim.get("tangents"), joint_indices=prim["joints"], joint_weights=prim["weights"])
    p = r3.Renderer(oh.LEFT, f32(16 / 9))
    o = OracleRenderer(oh.LEFT, f32(16 / 9))
    pm, om = p.add_mesh(prim["positions"], idx, **kw), o.add_mesh(prim["positions"], idx, **kw)
    assert p.meshes[pm].attr_off == o.meshes[om].attr_off and p.meshes[pm].joint_off == o.meshes[om].joint_off
    rng = np.random.Generator(np.random.PCG64(0x5141))
    ang = rng.uniform(-1.2, 1.2, (n, 2, 3)).astype(f32)
    poses = np.tile(oh.identity(), (n, 2, 1)).astype(f32)
    cx, sx, cz, sz = np.cos(ang[..., 0]), np.sin(ang[..., 0]), np.cos(ang[..., 2]), np.sin(ang[..., 2])
    sc = rng.uniform(0.5, 2.0, (n, 2, 3)).astype(f32)
    # Rz(a2) * Rx(a0) * S, column-major, plus a translation
    poses[..., 0], poses[..., 1], poses[..., 2] = cz * sc[..., 0], sz * sc[..., 0], 0.0
    poses[..., 4], poses[..., 5], poses[..., 6] = -sz * cx * sc[..., 1], cz * cx * sc[..., 1], sx * sc[..., 1]
    poses[..., 8], poses[..., 9], poses[..., 10] = sz * sx * sc[..., 2], -cz * sx * sc[..., 2], cx * sc[..., 2]
    poses[..., 12:15] = rng.uniform(-0.5, 0.5, (n, 2, 3)).astype(f32)
    base_words = p.mesh_cursor
    assert base_words == len(o.mesh_words)
    sks = p.add_skeletons_bulk(pm, list(poses))
    sk_in, sk_m = p.skinning_buffers()
    assert len(sk_in) == n and int(sk_in[:, 9].sum()) == 8_000_000 and np.array_equal(sk_m.reshape(n, 2, 16), poses)
    p._check(p.lib.r3n_skinning(p.ctx, r3._ffi.ptr(sk_in), n, r3._ffi.ptr(sk_m), len(sk_m)), "r3n_skinning")
    tail = p.mesh_cursor - base_words
    got = p.readback_mesh_words(4 * base_words, tail)
    # the oracle over the same records: its mesh buffer = the same base mesh + a zeroed tail of the same size
    words = np.concatenate([o.mesh_words, np.zeros(tail, dtype=np.uint32)])
    o.lib.r3o_skinning(o.lib.ptr(words), o.lib.ptr(sk_in), n, o.lib.ptr(sk_m))
    want = words[base_words:]
    assert want.any() and np.array_equal(got, want), f"{int((got != want).sum())} of {tail} skinned words differ"
    # and drawn: a few instances as objects through the whole frame
    mat = p.add_material(r3.material_record(albedo=(0.7, 0.7, 0.7, 1), albedo_mode="value", roughness=0.6), 0)
    for i in range(0, n, 5000):
        p.add_object(None, mat, oh.translation((-9.0 + 0.0004 * i, 0.0, 0.0)), skeleton=sks[i])
    p.set_camera_data(oh.mat4_mul(oh.from_euler_xyz(0.0, 0.0, 0.0), oh.translation((0.0, 0.0, 12.0))), ("perspective", 60.0, 0.1))
    out = p.render(640, 360)
    assert out["pass"].sum() > 0 and (out["vis"] != 0).sum() > 100
    assert np.array_equal(p.readback_mesh_words(4 * base_words, tail), want)  # the frame's own skinning pass: same result
    p.close()


def test_oracle_mfma_order_is_close_to_the_contract():
    """The FMA-ordered restatement (what the matrix-core kernel computes) against the contract's order on posed rigs: same
    values to rounding -- a handful of ulps -- so choosing the kernel is a performance decision, not a visual one."""
    for joints in (2, 4):
        o = OracleRenderer(oh.LEFT)
        _mesh, _mat, sks, _pos, _nrm = _rig(o, joints, omk, n_skeletons=2)
        for i, sk in enumerate(sks):
            o.set_skeleton_joint_matrices(sk, _pose(joints, 40 + i))
        sk_in, sk_m = o.skinning_buffers()
        o.lib.r3o_skinning(o.lib.ptr(o.mesh_words), o.lib.ptr(sk_in), len(sk_in), o.lib.ptr(sk_m))
        exact = [a.view(f32).copy() for sk in sks for a in _skinned_runs(o, sk)]
        nj = np.full(len(sk_in), joints, dtype=np.uint32)
        o.lib.r3o_skinning_mfma_order(o.lib.ptr(o.mesh_words), o.lib.ptr(sk_in), len(sk_in), o.lib.ptr(sk_m), o.lib.ptr(nj))
        fused = [a.view(f32).copy() for sk in sks for a in _skinned_runs(o, sk)]
        for a, b in zip(exact, fused):
            assert np.allclose(a, b, rtol=2e-6, atol=2e-6)
        assert any((a != b).any() for a, b in zip(exact, fused)), "the two operation orders never differ: is the fused path wired up?"


@pytest.mark.gpu
@pytest.mark.parametrize("joints", [2, 4])
def test_gpu_skinning_mfma_matches_its_oracle_order(joints):
    """R3N_SKIN_MFMA (v_mfma_f32_16x16x4_f32: four joint matrices x 16 vertices per instruction) is bit-identical to the
    FMA-ordered oracle restatement on rigs of up to four joints -- several skeletons, ragged last wave, the zero-weight and
    four-influence vertices of the test rig -- and refuses larger rigs loudly."""
    import torch
    assert torch.cuda.is_available()
    import rend3_amd as r3
    o, p = OracleRenderer(oh.LEFT, f32(1.5)), r3.Renderer(oh.LEFT, f32(1.5))
    rigs = []
    for r, mk in ((o, omk), (p, r3.material_record)):
        _mesh, mat, sks, _pos, _nrm = _rig(r, joints, mk, n_skeletons=5)
        for i, sk in enumerate(sks):
            r.add_object(None, mat, oh.translation((-2.4 + 1.2 * i, 0.0, 0.0)), skeleton=sk)
            r.set_skeleton_joint_matrices(sk, _pose(joints, 900 + i))
        r.set_camera_data(oh.look_at_lh((0, 1.2, -4), (0, 1, 0), (0, 1, 0)), ("perspective", 60.0, 0.1))
        rigs.append(sks)
    p.set_skinning_mode(1)
    p.render(96, 64, readback=False)
    sk_in, sk_m = o.skinning_buffers()
    nj = np.full(len(sk_in), joints, dtype=np.uint32)
    o.lib.r3o_skinning_mfma_order(o.lib.ptr(o.mesh_words), o.lib.ptr(sk_in), len(sk_in), o.lib.ptr(sk_m), o.lib.ptr(nj))
    for i in range(5):
        for a, b in zip(_skinned_runs(o, rigs[0][i]), _skinned_runs(p, rigs[1][i])):
            bad = np.nonzero(a != b)[0]
            assert len(bad) == 0, (i, len(bad), a.view(f32)[bad[:4]], b.view(f32)[bad[:4]])
    # a rig with more than four joints is refused in this mode
    q = r3.Renderer(oh.LEFT, f32(1.5))
    _mesh, mat, (sk,), _pos, _nrm = _rig(q, 7, r3.material_record)
    q.add_object(None, mat, oh.identity(), skeleton=sk)
    q.set_skinning_mode(1)
    with pytest.raises(Exception, match="four joints"):
        q.render(64, 64, readback=False)
    q.close()
    p.close()
