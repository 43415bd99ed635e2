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


def test_group_eight_entries_vit_l_shapes(api, pkg, tmp_path):
    """The shape of the 8-GPU node on the one GPU there is: an 8-entry device list (all device 0) at ViT-L/14 widths (2 layers),
    global batches 13 (B % G != 0: shards of 2 and 1 images), 3 (B < G: five entries idle) and 8, with TWO jobs in flight.  Every
    output equals one session's result bit for bit, whatever the number of entries."""
    path = str(tmp_path / "large2.gguf")
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=13, layers=2)
    imgs = pkg.synth.synthetic_images(13, 224, 224, seed=13)
    sess = api.Session(api.Model(path, classify=True))
    ref = sess.predict(imgs, classify=True, topk=5)
    grp = api.Group(path, devices=[0] * 8, classify=True)
    assert grp.size == 8
    got = grp.predict(imgs, classify=True, topk=5)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    # two jobs in flight: B = 3 behind B = 13, then B = 8 behind B = 3
    h13 = grp.submit(imgs, classify=True)
    h3 = grp.submit(imgs[5:8], classify=False)
    o13 = grp.wait(h13)
    h8 = grp.submit(imgs[2:10], classify=True)
    o3 = grp.wait(h3)
    o8 = grp.wait(h8)
    assert np.array_equal(o13["logits"], ref["logits"])
    f3 = sess.predict(imgs[5:8], classify=False)
    for k in f3:
        assert np.array_equal(o3[k], f3[k]), k
    assert np.array_equal(o8["logits"], ref["logits"][2:10]) and np.array_equal(o8["patch_tokens"], ref["patch_tokens"][2:10])
    grp.close()


def test_group_predict_refused_while_a_ticket_is_pending(api, golden_dir):
    """dinov2_hip_group_predict = submit + wait as one unit: with an un-waited ticket ahead of it the call is refused BEFORE anything is
    queued (round 3: the job was queued, its wait refused, and the orphan wrote into the caller's buffers after the error return).
    The pending ticket stays waitable and the group usable."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    grp = api.Group(gguf, devices=[0, 0], classify=True)
    imgs = np.random.default_rng(21).standard_normal((5, 3, 56, 84)).astype(np.float32)
    ref = sess.predict(imgs, classify=True)
    h = grp.submit(imgs, classify=True)
    with pytest.raises(api.DinoError) as e:
        grp.predict(imgs[:2], classify=True)
    assert e.value.status == 4 and "not waited" in str(e.value)
    assert np.array_equal(grp.wait(h)["logits"], ref["logits"])
    assert np.array_equal(grp.predict(imgs[:2], classify=True)["logits"], ref["logits"][:2])
    grp.close()


def test_group_rejects_bad_descriptors_before_copying(api, golden_dir):
    """Layout / height / width are validated in dinov2_hip_group_submit, before turnstile 0 sizes a host -> device copy from them: an
    unknown layout, a size that is not a multiple of the patch size and a zero batch come back as INVALID with nothing queued (the
    next call runs as ticket 0 would)."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    grp = api.Group(gguf, devices=[0, 0], classify=True)
    L = api.lib()
    buf = np.zeros((2, 3, 56, 84), np.float32)
    err = C.create_string_buffer(256)
    t = C.c_int64(-1)
    for layout, hh, ww, b in ((7, 56, 84, 2), (api.RGB_CHW, 57, 84, 2), (api.RGB_CHW, 56, 84, 0), (api.RGB_CHW, 1 << 20, 84, 2)):
        if (hh, ww, b) == (1 << 20, 84, 2):
            hh = (1 << 20) + 1  # not a multiple of 14: rejected before any size arithmetic is trusted
        i = api.Input(buf.ctypes.data, b, hh, ww, layout, 0)
        assert L.dinov2_hip_group_submit(grp._h, C.byref(i), None, api.CLASSIFY, C.byref(t), err, len(err)) == 4, (layout, hh, ww, b)
        assert t.value == -1
    ref = api.Session(api.Model(gguf, classify=True)).predict(buf, classify=True)
    assert np.array_equal(grp.predict(buf, classify=True)["logits"], ref["logits"])
    grp.close()


def test_group_shard_cut_into_passes(api, golden_dir, monkeypatch):
    """A shard longer than one pass of the forward takes (forced here with DINOV2_HIP_MAX_CHUNK = 2) goes through the group with ONE
    chunked predict inside the forward turnstile, results leaving pass by pass; same bits as the un-chunked single session, for one
    and for two jobs in flight."""
    gguf = os.path.join(golden_dir, "tiny_swiglu_reg4.gguf")
    imgs = np.random.default_rng(31).standard_normal((11, 3, 70, 84)).astype(np.float32)
    ref = api.Session(api.Model(gguf, classify=True)).predict(imgs, classify=True, topk=2)
    grp = api.Group(gguf, devices=[0, 0], classify=True)
    monkeypatch.setenv("DINOV2_HIP_MAX_CHUNK", "2")  # shards of 6 and 5 images -> 3 passes each
    got = grp.predict(imgs, classify=True, topk=2)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    h1 = grp.submit(imgs, classify=True)
    h2 = grp.submit(imgs[:3], classify=True)  # 2 + 1 images: the first entry's shard is exactly one pass
    assert np.array_equal(grp.wait(h1)["patch_tokens"], ref["patch_tokens"])
    assert np.array_equal(grp.wait(h2)["logits"], ref["logits"][:3])
    monkeypatch.delenv("DINOV2_HIP_MAX_CHUNK")
    assert np.array_equal(grp.predict(imgs, classify=True)["logits"], ref["logits"])
    grp.close()


def test_group_predict_only_callers_stay_on_one_lane(api, pkg, tmp_path):
    """With the default two lanes per device a caller that only uses the blocking dinov2_hip_group_predict has one job in flight: it
    must keep running on lane 0, so that the second lane never allocates a workspace (round 3 alternated lanes by ticket parity and
    doubled the device memory of every predict-only user).  Device memory in use after a few calls: two-lane group == one-lane group."""
    import torch
    path = str(tmp_path / "large2.gguf")
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=5, layers=2)
    imgs = pkg.synth.synthetic_images(4, 518, 518, seed=5)

    def used_after(lanes):
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        grp = api.Group(path, devices=[0], classify=True, streams_per_device=lanes, broadcast=False)
        for _ in range(4):
            out = grp.predict(imgs, classify=True)
        free1, _ = torch.cuda.mem_get_info()
        grp.close()
        return free0 - free1, out

    one, o1 = used_after(1)
    two, o2 = used_after(2)
    assert np.array_equal(o1["logits"], o2["logits"])
    ws = api.Model(path, classify=True).workspace_bytes(4, 518, 518)
    assert ws > 64 << 20  # the workspace is what would be doubled: large enough to stand out
    assert two < one + ws // 2, (one, two, ws)


def test_fetch_refused_after_the_workspace_was_overwritten(api, golden_dir):
    """dinov2_hip_fetch copies out the LAST predict: dinov2_hip_debug_hidden overwrites the workspace and a failed predict may have
    re-carved it -- after either, fetch must say INVALID instead of returning rows of whatever is there now."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    imgs = np.random.default_rng(9).standard_normal((2, 3, 56, 84)).astype(np.float32)
    L = api.lib()
    err = C.create_string_buffer(256)
    out, o = api._alloc_outputs(sess.model.hparams, 2, 56, 84, api.RGB_CHW, True, 0, ("cls", "patch_tokens", "logits", "probs"))
    i = api.Input(imgs.ctypes.data, 2, 56, 84, api.RGB_CHW, 0)
    assert L.dinov2_hip_predict(sess._h, C.byref(i), None, api.CLASSIFY, err, len(err)) == 0
    assert L.dinov2_hip_fetch(sess._h, C.byref(o), err, len(err)) == 0
    sess.debug_hidden(imgs, 1)
    assert L.dinov2_hip_fetch(sess._h, C.byref(o), err, len(err)) == 4
    assert L.dinov2_hip_predict(sess._h, C.byref(i), None, api.CLASSIFY, err, len(err)) == 0
    bad = api.Input(0, 2, 56, 84, api.RGB_CHW, 0)  # null data: fails in the argument checks, before anything ran
    assert L.dinov2_hip_predict(sess._h, C.byref(bad), None, api.CLASSIFY, err, len(err)) == 4
    assert L.dinov2_hip_fetch(sess._h, C.byref(o), err, len(err)) == 0  # the workspace was not touched: the last forward is still there


def test_group_predict_races_a_submitting_thread(api, golden_dir):
    """dinov2_hip_group_predict used to check "no un-waited ticket" in one critical section and enqueue in another, so a
    submit from a second thread could slip in between and orphan predict's ticket (a job nobody could wait for, writing into the caller's
    buffers).  Two threads hammer one group -- A calls the blocking predict, B keeps one ticket in flight with submit / wait -- for a few
    hundred rounds: every call either succeeds with the single-session bits or is REFUSED with the documented error; nothing hangs, and
    when both are done the pipeline is empty (a plain predict succeeds)."""
    import threading
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    grp = api.Group(gguf, devices=[0, 0], classify=True, streams_per_device=2)
    sess = api.Session(api.Model(gguf, classify=True))
    rng = np.random.default_rng(21)
    a_img, b_img = (rng.standard_normal((3, 3, 56, 84)).astype(np.float32) for _ in range(2))
    ref_a, ref_b = sess.predict(a_img, classify=True)["logits"], sess.predict(b_img, classify=True)["logits"]
    stats = {"a_ok": 0, "a_refused": 0, "b_ok": 0, "b_refused": 0, "bad": []}
    stop = threading.Event()

    def thread_a():
        for _ in range(300):
            try:
                out = grp.predict(a_img, classify=True, want=("logits",))
                if not np.array_equal(out["logits"], ref_a):
                    stats["bad"].append("A: wrong bits")
                stats["a_ok"] += 1
            except api.DinoError as e:
                if "not waited for yet" not in str(e) and "in flight" not in str(e) and "submission order" not in str(e):
                    stats["bad"].append("A: " + str(e))
                stats["a_refused"] += 1
        stop.set()

    def thread_b():
        while not stop.is_set():
            try:
                h = grp.submit(b_img, classify=True, want=("logits",))
            except api.DinoError as e:
                if "in flight" not in str(e):
                    stats["bad"].append("B submit: " + str(e))
                stats["b_refused"] += 1
                continue
            while True:  # a ticket taken by B must be waitable by B: predict's own ticket may be ahead of it for a moment
                try:
                    out = grp.wait(h)
                    break
                except api.DinoError as e:
                    if "submission order" not in str(e):
                        stats["bad"].append("B wait: " + str(e))
                        return
            if not np.array_equal(out["logits"], ref_b):
                stats["bad"].append("B: wrong bits")
            stats["b_ok"] += 1

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start()
    ta.join(timeout=300); stop.set(); tb.join(timeout=60)
    assert not ta.is_alive() and not tb.is_alive(), "a call hung"
    assert not stats["bad"], stats["bad"][:5]
    assert stats["a_ok"] + stats["a_refused"] == 300 and stats["b_ok"] > 0
    assert np.array_equal(grp.predict(a_img, classify=True, want=("logits",))["logits"], ref_a)  # the pipeline is empty and usable
    grp.close()
