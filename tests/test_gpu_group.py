"""dinov2_hip_group_* (SURVEY 8(e)): the native multi-device driver behind the C-ABI -- one host thread + session per device,
contiguous batch split, outputs written at the shard offsets of the caller's buffers, one-time RCCL broadcast of the weight
arena.  On a 1-GPU box the split logic runs with the SAME device listed twice (every entry then reads the file itself) and
the RCCL path with a one-device communicator; with >= 2 visible devices the real two-device broadcast runs too."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ndev():
    import torch
    return torch.cuda.device_count()


def test_group_split_matches_single_session(api, golden_dir):
    """Two ranks on device 0, ragged global batch 7 (4 + 3), classify and features, f32 and raw 8-bit input: every output equals
    the single-session result bit for bit (B images are B independent forwards)."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    grp = api.Group(gguf, devices=[0, 0], classify=True)
    assert grp.size == 2 and grp.broadcast_ms < 0  # duplicate device: no communicator, both entries read the file
    sess = api.Session(api.Model(gguf, classify=True))
    rng = np.random.default_rng(7)
    imgs = rng.standard_normal((7, 3, 70, 98)).astype(np.float32)
    for classify in (True, False):
        ref = sess.predict(imgs, classify=classify, topk=3 if classify else 0)
        got = grp.predict(imgs, classify=classify, topk=3 if classify else 0)
        assert ref.keys() == got.keys()
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (classify, k)
    raw = rng.integers(0, 256, (5, 61, 83, 3), dtype=np.uint8)
    a = sess.predict(raw, classify=False, layout=api.U8_BGR_HWC, want=("patch_tokens",))
    b = grp.predict(raw, classify=False, layout=api.U8_BGR_HWC, want=("patch_tokens",))
    assert np.array_equal(a["patch_tokens"], b["patch_tokens"])
    one = grp.predict(imgs[:1], classify=True)  # B < G: the second rank idles
    assert np.array_equal(one["logits"], sess.predict(imgs[:1], classify=True)["logits"])
    with pytest.raises(api.DinoError) as e:
        grp.predict(np.zeros((2, 3, 60, 70), np.float32))
    assert e.value.status == 4
    grp.close()


def test_group_vit_l_shapes_equal_single_session(api, pkg, tmp_path):
    """ViT-L/14 shapes (2 layers), global batch 2 over two ranks -> every rank computes ONE image with the few-tile GEMM plans.  The
    group's result equals one session's batch-2 result bit for bit, i.e. it does not depend on the number of devices; a single CHW
    image is promoted to a batch of one like Session.predict does."""
    path = str(tmp_path / "large2.gguf")
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=11, layers=2)
    imgs = pkg.synth.synthetic_images(2, 518, 518, seed=11)
    ref = api.Session(api.Model(path, classify=True)).predict(imgs, classify=True)
    grp = api.Group(path, devices=[0, 0], classify=True)
    got = grp.predict(imgs, classify=True)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    single = grp.predict(imgs[1], classify=True)  # [3, H, W]
    assert single["logits"].shape == (1, 1000) and np.array_equal(single["logits"][0], ref["logits"][1])
    with pytest.raises(ValueError):
        grp.predict(np.zeros((2, 5, 518, 518), np.float32), classify=True)
    grp.close()


def test_group_pipelined_jobs_equal_plain_predict(api, golden_dir):
    """dinov2_hip_group_submit / _wait with streams_per_device = 1, 2, 3 jobs in flight (copy-in, forward and copy-out of
    consecutive batches overlapping across a device's lanes), pageable and page-locked host buffers, ragged batches incl. B <
    devices, classify and features: every output bit for bit what a plain single-session predict returns; the in-flight limit
    and the wait order are enforced."""
    gguf = os.path.join(golden_dir, "tiny_swiglu_reg4.gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    rng = np.random.default_rng(3)
    batches = [rng.standard_normal((b, 3, 84, 70)).astype(np.float32) for b in (9, 5, 1, 4, 7, 2)]
    pinned = []
    for x in batches:
        p = api.pinned_empty(x.shape, np.float32)
        p[...] = x
        pinned.append(p)
    for lanes in (1, 2, 3):
        grp = api.Group(gguf, devices=[0, 0], classify=True, streams_per_device=lanes)
        for classify in (True, False):
            refs = [sess.predict(x, classify=classify, topk=2 if classify else 0) for x in batches]
            for src in (batches, pinned):
                inflight, got = [], []
                for x in src:
                    if len(inflight) == lanes:
                        got.append(grp.wait(inflight.pop(0)))
                    inflight.append(grp.submit(x, classify=classify, topk=2 if classify else 0))
                while inflight:
                    got.append(grp.wait(inflight.pop(0)))
                for r, o in zip(refs, got):
                    assert r.keys() == o.keys()
                    for k in r:
                        assert np.array_equal(r[k], o[k]), (lanes, classify, k)
        hs = [grp.submit(batches[0], classify=True) for _ in range(lanes)]
        with pytest.raises(api.DinoError):  # lanes + 1 jobs in flight
            grp.submit(batches[0], classify=True)
        if lanes > 1:
            with pytest.raises(api.DinoError):  # out of order
                grp.wait(hs[1])
        for hnd in hs:
            assert np.array_equal(grp.wait(hnd)["logits"], sess.predict(batches[0], classify=True)["logits"])
        assert np.array_equal(grp.predict(pinned[1], classify=True)["logits"], sess.predict(batches[1], classify=True)["logits"])
        grp.close()


def test_fetch_after_forward_only_predict(api, golden_dir):
    """dinov2_hip_fetch: predict with out = NULL (forward only), then the copy-out as its own call -- same bits as the one-call form."""
    import ctypes as C
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    imgs = np.random.default_rng(5).standard_normal((3, 3, 56, 84)).astype(np.float32)
    ref = sess.predict(imgs, classify=True, topk=3)
    L = api.lib()
    err = C.create_string_buffer(256)
    out, o = api._alloc_outputs(sess.model.hparams, 3, 56, 84, api.RGB_CHW, True, 3, ("cls", "patch_tokens", "logits", "probs"))
    assert L.dinov2_hip_fetch(api.Session(sess.model)._h, C.byref(o), err, len(err)) == 4  # nothing to fetch yet
    i = api.Input(imgs.ctypes.data, 3, 56, 84, api.RGB_CHW, 0)
    assert L.dinov2_hip_predict(sess._h, C.byref(i), None, api.CLASSIFY, err, len(err)) == 0
    assert L.dinov2_hip_fetch(sess._h, C.byref(o), err, len(err)) == 0, err.value
    for k in ref:
        assert np.array_equal(ref[k], out[k]), k


def test_group_rccl_broadcast_one_rank(api, golden_dir):
    """broadcast = 1 with a single device: dlopen(librccl), ncclCommInitAll, the arena broadcast (root to itself) and teardown all
    run; results equal the plain session's."""
    gguf = os.path.join(golden_dir, "tiny_swiglu_reg4.gguf")
    grp = api.Group(gguf, devices=[0], classify=True, broadcast=True)
    assert grp.size == 1 and grp.broadcast_ms >= 0
    imgs = np.random.default_rng(1).standard_normal((3, 3, 56, 84)).astype(np.float32)
    ref = api.Session(api.Model(gguf, classify=True)).predict(imgs, classify=True)
    got = grp.predict(imgs, classify=True)
    assert np.array_equal(ref["logits"], got["logits"])


def test_group_two_devices_broadcast(api, golden_dir):
    """World = 2 when two devices are visible (skipped on the 1-GPU box): rank 1 never reads tensor data from the file -- its
    arena arrives by ncclBroadcast over xGMI -- and its half of the batch must still equal device 0's single-session result."""
    if _ndev() < 2:
        pytest.skip("needs >= 2 visible devices")
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    grp = api.Group(gguf, devices=[0, 1], classify=True, broadcast=True)
    assert grp.size == 2 and grp.broadcast_ms >= 0
    imgs = np.random.default_rng(2).standard_normal((6, 3, 70, 70)).astype(np.float32)
    ref = api.Session(api.Model(gguf, classify=True)).predict(imgs, classify=True)
    got = grp.predict(imgs, classify=True)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k


def test_group_from_cpp(tmp_path, golden_dir):
    """A C++ host (the reference's mains are C++) drives the group through include/dinov2_hip.h alone."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "group_smoke")
    libdir = os.path.join(root, "dinov2.cpp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "group_smoke.cpp"),
                           "-o", exe, "-L" + libdir, "-ldinov2_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, os.path.join(golden_dir, "tiny_gelu_reg4.gguf")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GROUP_OK" in r.stdout, r.stdout + r.stderr
