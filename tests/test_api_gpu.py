"""Reference-API behaviour of the mirrors on the MI355X: caching, checkpoints, loaders, harness."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def small_clip(seed=3, **kw):
    from eventclip_amd import clip as eclip
    cfg = eclip.arch_config('ViT-B/32', layers=2, text_layers=2, vocab_size=49408)
    sd = eclip.random_state_dict(cfg, seed=seed)
    return cfg, sd, eclip.CLIP(cfg, sd, **kw).cuda().eval()


def test_text_feats_cache_and_explicit_class_names(hip):
    """clip_cls.py:64-93: cached when called without names or with matching names; explicit,
    different names are computed but not cached (and do not crash as upstream does)."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    cfg, sd, m = small_clip()
    names = ['a', 'b', 'c']
    tokens = eclip.synthetic_tokens(3, seed=0)
    model = ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a {}', class_names=names,
                                            agg_func='sum', class_tokens=tokens)).cuda().eval()
    assert model.text_feats is None
    t1 = model.get_text_feats()
    assert model.text_feats is t1 and model.get_text_feats() is t1
    assert model.get_text_feats(names) is t1
    torch.testing.assert_close(t1.norm(dim=-1), torch.ones(3, device='cuda'), rtol=1e-5, atol=1e-5)
    assert abs(model.logit_scale - 100.0) < 1e-3 and model.dtype == torch.float32
    assert model.train() is model and not m.training             # CLIP stays in eval (:202-206)
    with pytest.raises(AssertionError):
        ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a {}', class_names=names,
                                        agg_func='median'))


def test_text_identity_adapter_and_checkpoint_round_trip(hip, tmp_path):
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import FSCLIPClassifier
    from oracle import classify as oc
    cfg, sd, m = small_clip(seed=4)
    K, B, T, C = 4, 3, 2, 512
    tokens = eclip.synthetic_tokens(K, seed=1)
    mk = lambda: FSCLIPClassifier(                                   # noqa: E731
        adapter_dict=dict(adapter_type='text-identity', in_dim=C, residual=False),
        clip_dict=dict(clip_model=m, prompt='a {}', class_names=list('wxyz'), agg_func='mean',
                       class_tokens=tokens),
        loss_dict=dict(use_logits_loss=True, use_probs_loss=False)).cuda().eval()
    model = mk()
    assert sorted(model.state_dict()) == ['adapter.dummy', 'text_feats']
    with torch.no_grad():
        model.text_feats.add_(torch.randn_like(model.text_feats) * 0.05)
    g = torch.Generator().manual_seed(0)
    valid = torch.tensor([[True, True], [True, False], [True, True]])
    imgs = torch.randn(B, T, 3, 224, 224, generator=g) * valid[:, :, None, None, None]
    out = model({'img': imgs.cuda(), 'valid_mask': valid.cuda()})
    feats = m.encode_image(imgs[valid].cuda()).cpu()
    full = torch.zeros(B, T, C)
    full[valid] = feats
    text = torch.nn.functional.normalize(model.text_feats.detach().cpu(), dim=-1)
    want = oc.fs_tail(full, valid, text, 100.0, 'mean')
    for k in ('full_logits', 'logits', 'probs'):
        torch.testing.assert_close(out[k].cpu(), want[k], rtol=1e-4, atol=1e-4)
    # nerv-style checkpoint: {'state_dict': ...} without any CLIP weight (clip_cls.py:208-219)
    path = os.path.join(tmp_path, 'model_1.pth')
    torch.save({'state_dict': model.state_dict()}, path)
    other = mk()
    assert not torch.equal(other.text_feats, model.text_feats)
    other.load_weight(path)
    out2 = other({'img': imgs.cuda(), 'valid_mask': valid.cuda()})
    assert torch.equal(out2['logits'], out['logits'])
    with pytest.raises(NotImplementedError):
        FSCLIPClassifier(adapter_dict=dict(adapter_type='mlp'),
                         clip_dict=dict(clip_model=m, prompt='a {}', class_names=list('wxyz'),
                                        agg_func='mean', class_tokens=tokens))


def test_clip_load_from_state_dict_file(hip, tmp_path):
    import torch
    from eventclip_amd import clip as eclip
    cfg, sd, m = small_clip(seed=5)
    path = os.path.join(tmp_path, 'ViT-B-32.pt')
    torch.save(sd, path)
    model, preprocess = eclip.load(path)
    assert model.visual.output_dim == 512 and model.cfg['layers'] == 2
    assert preprocess.n_px == 224
    img = torch.randn(2, 3, 224, 224).cuda()
    assert torch.equal(model.encode_image(img), m.encode_image(img))
    model2, _ = eclip.load('ViT-B/32', download_root=str(tmp_path))   # arch name -> <root>/ViT-B-32.pt
    assert torch.equal(model2.encode_image(img), m.encode_image(img))
    li, lt = model(img, eclip.synthetic_tokens(3, seed=2).cuda())      # CLIP.forward
    assert tuple(li.shape) == (2, 3) and torch.equal(li.t(), lt)


def test_evaluate_harness_with_pipeline(hip):
    """test.py:55-93 meters over two uneven batches of raw events."""
    import torch
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.harness import evaluate
    from eventclip_amd.synthetic import make_batch
    cfg, sd, m = small_clip(seed=6)
    K = 6
    model = ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a {}',
                                            class_names=[str(i) for i in range(K)], agg_func='mean',
                                            class_tokens=eclip.synthetic_tokens(K, seed=3))).cuda().eval()
    qa = dict(max_imgs=2, N=30000, split_method='event_count', convert_method='event_histogram',
              grayscale=True, count_non_zero=True, background_mask=False)
    pipe = Event2ImagePipeline((100, 120), 60000, qa, n_px=224, patch=32, kpad=m.kpad)
    b1, b2 = make_batch(3, 40000, (100, 120), seed=1), make_batch(1, 12500, (100, 120), seed=2)
    preds = [model(pipe(b))['logits'].argmax(-1).cpu() for b in (b1, b2)]
    labels = [preds[0].clone(), (preds[1] + 1) % K]                    # first batch right, second wrong
    accs = evaluate(model, [dict(events=b1, label=labels[0]), dict(events=b2, label=labels[1])],
                    is_nin=True, pipeline=pipe)
    assert abs(accs['logits_acc'] - 0.75) < 1e-6                       # (1.0 * 3 + 0.0 * 1) / 4
    assert set(accs) == {'probs_acc', 'logits_acc', 'probs_acc5', 'logits_acc5'}
    assert accs['logits_acc5'] >= accs['logits_acc']


def test_views_match_the_references_event2image_dataset(hip):
    """Frames, zero padding, masks, the random chunk subset (same torch RNG stream) and the four TTA views
    in order, against the reference's own Event2ImageDataset run with an identity transform
    (tests/golden/event2img.npz)."""
    import os
    import numpy as np
    import torch
    from conftest import GOLDEN
    from eventclip_amd.event2img import Event2ImagePipeline
    z = np.load(os.path.join(GOLDEN, 'event2img.npz'))
    qa = dict(max_imgs=int(z['qa_max_imgs']), split_method=str(z['qa_split_method']),
              convert_method=str(z['qa_convert_method']), N=int(z['qa_N']), grayscale=bool(z['qa_grayscale']),
              count_non_zero=bool(z['qa_count_non_zero']), background_mask=bool(z['qa_background_mask']))
    res = tuple(int(v) for v in z['resolution'])
    pipe = Event2ImagePipeline(res, int(z['max_n']), qa, n_px=224)
    T = pipe.max_imgs

    def views(ev_d, n, hflip, tflip):
        fr, ri, vm = pipe.plan([n], tflip=tflip)
        frames = pipe.frames(ev_d, fr.cuda(), hflip=hflip, tflip=tflip).cpu().numpy()
        full = np.zeros((T, *res, 3), dtype=np.uint8)                      # padded views are all-zero tensors
        for t in range(T):
            if ri[0, t] >= 0:
                full[t] = frames[int(ri[0, t])]
        return full, vm[0].numpy()

    for i, n in enumerate(z['counts']):
        ev_d = torch.from_numpy(z[f'events{i}']).cuda()
        torch.manual_seed(1000 + i)
        got, vm = views(ev_d, int(n), False, False)
        np.testing.assert_array_equal(vm, z[f'tta0_valid{i}'])
        np.testing.assert_array_equal(got, z[f'tta0_img{i}'].transpose(0, 2, 3, 1))
        torch.manual_seed(1000 + i)
        for v, (h, t) in enumerate(((False, False), (True, False), (False, True), (True, True))):
            got, vm = views(ev_d, int(n), h, t)                                # event2img.py:97-103 order
            np.testing.assert_array_equal(vm, z[f'tta1_valid{i}'][v])
            np.testing.assert_array_equal(got, z[f'tta1_img{i}'][v].transpose(0, 2, 3, 1), err_msg=f'sample {i} view {v}')


def test_classifier_forward_runs_through_the_registered_custom_ops(hip):
    """The drop-in classes reach the kernels through torch.ops.eventclip_hip.* (north_star's boundary):
    chaining the ops by hand gives the classifier's out_dict bit for bit, and every op is seen by a
    torch dispatch trace of model.forward."""
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    from eventclip_amd import _lib, torch_ops, vis
    from eventclip_amd import clip as eclip
    from eventclip_amd.clip_cls import ZSCLIPClassifier
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import make_batch
    qa = dict(max_imgs=3, N=20000, split_method='event_count', convert_method='event_histogram',
              grayscale=False, count_non_zero=False, background_mask=True)
    cfg = eclip.arch_config('ViT-B/32', layers=2, text_layers=1)
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, seed=3)).cuda().eval()
    tokens = eclip.synthetic_tokens(7, seed=1)
    model = ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a {}', class_names=list('abcdefg'),
                                            agg_func='mean', class_tokens=tokens)).cuda().eval()
    pipe = Event2ImagePipeline((180, 240), 225000, qa, n_px=224, patch=32, kpad=m.kpad)
    pipe.strict = False
    evs = make_batch(2, [50000, 21000], (180, 240), seed=4)

    seen = []

    class Trace(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if 'eventclip_hip' in str(func):
                seen.append(str(func).split('.')[1])
            return func(*args, **(kwargs or {}))

    with Trace():
        out = model(pipe(evs))
    assert {'events_to_frames', 'preprocess', 'vit_encode', 'text_encode', 'classify'} <= set(seen), seen

    # the same chain by hand
    ev, n_ev = pipe._concat(evs, torch.device('cuda'))
    fr, ri, vm = pipe.plan(n_ev)
    ops = torch.ops.eventclip_hip
    frames = ops.events_to_frames(ev, fr.cuda(), 180, 240, 10., [255, 0, 0], [0, 0, 255], False, True, 20000,
                                  False, False, False, 0)
    patches = ops.preprocess(frames, 224, _lib.EC_PRE_PATCHES16, 32, m.kpad, _lib.EC_F16)
    feats = ops.vit_encode(patches, torch_ops.handle_of(m))
    from eventclip_amd.clip_cls import _l2_normalize
    text = _l2_normalize(ops.text_encode(tokens.cuda().int(), torch_ops.handle_of(m)))
    full, logits, probs = ops.classify(feats, ri.cuda(), text.t().contiguous(), float(model.logit_scale), 1, False)
    assert torch.equal(full, out['full_logits']) and torch.equal(logits, out['logits'])
    assert torch.equal(probs, out['probs'])


def test_host_feeder_is_bit_identical_to_the_synchronous_path(hip):
    """Event2ImagePipeline.stream (pinned staging ring, copy stream, batch i + 1 uploaded while the GPU works on
    batch i) against pipe(list_of_arrays) (np.concatenate + synchronous copy): the same patches, masks and view
    order for every batch, float and packed events, ragged batch sizes, more batches than ring slots."""
    import torch
    from eventclip_amd import vis
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import GEOMETRY, make_events
    g = GEOMETRY['n_caltech']
    qa = dict(max_imgs=10, N=g['N'], split_method='event_count', convert_method='event_histogram', grayscale=False,
              count_non_zero=False, background_mask=True)
    pipe = Event2ImagePipeline(g['resolution'], g['max_n'], qa, n_px=224, patch=32, kpad=6144)
    sizes = [[50000, 20000, 93000], [200000], [7000, 41000], [120000, 60000, 20000, 20001], [30000]]
    batches = [[make_events(n, g['resolution'], seed=10 * b + i) for i, n in enumerate(ns)] for b, ns in enumerate(sizes)]
    for packed in (False, True):
        bs = [[vis.pack_events(e) for e in b] for b in batches] if packed else batches
        want = [pipe(b) for b in bs]
        got = list(pipe.stream(iter(bs), depth=2, copy_threads=3))
        assert len(got) == len(want)
        for a, b in zip(got, want):
            for k in ('patches', 'valid_mask', 'row_idx'):
                assert torch.equal(a[k], b[k]), (packed, k)
    # harness-style dicts keep their other entries
    dd = [dict(events=b, label=torch.tensor([1] * len(b))) for b in batches[:2]]
    out = list(pipe.stream(iter(dd)))
    assert all('label' in o and 'events' not in o for o in out)
    # an exception in the producer thread reaches the caller
    import pytest
    with pytest.raises(IndexError):
        list(pipe.stream(iter([[make_events(0, g['resolution'])]])))


def test_host_feeder_dropped_while_the_ring_regrows_ends_its_thread(hip):
    """the wait for ALL slots that precedes a ring re-allocation (a batch larger than any before it) ran in
    HostFeeder._parse under a strong reference to the feeder and without a timeout -- a consumer that dropped the
    feeder at that moment left the thread, the pinned ring and the device ring alive for ever.  The wait now runs in
    the producer loop on the queues alone: a regrown ring still delivers bit-identical batches, and a feeder dropped
    while the producer waits for the slots is collected and its thread ends."""
    import gc
    import time
    import weakref
    import torch
    from eventclip_amd.event2img import Event2ImagePipeline, HostFeeder
    from eventclip_amd.synthetic import GEOMETRY, make_batch
    g = GEOMETRY['n_cars']
    qa = dict(max_imgs=4, N=g['N'], split_method='event_count', convert_method='event_histogram', grayscale=True,
              count_non_zero=True, background_mask=False)
    pipe = Event2ImagePipeline(g["resolution"], 4 * g["N"], qa, n_px=224, patch=32, kpad=6144)   # 4 views: no random view subsampling
    assert pipe.max_imgs == 4
    small = [make_batch(2, 4000, g['resolution'], seed=i) for i in range(3)]
    big = make_batch(2, 90000, g['resolution'], seed=9)                    # 2.9 MB against a 1 MB ring: regrow
    feeder = HostFeeder(pipe, small + [big] + small[:1], depth=2)
    outs = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in feeder]
    assert len(outs) == 5 and feeder.capacity >= 2 * 90000 * 16
    for o, batch in zip(outs, small + [big] + small[:1]):
        want = pipe(batch)
        assert torch.equal(o['patches'], want['patches']) and torch.equal(o['valid_mask'], want['valid_mask'])
    # dropped while the producer waits for every slot (the consumer holds one: it never took the first batch's slot back)
    feeder = HostFeeder(pipe, small[:2] + [big], depth=2)
    next(feeder)                                       # one batch consumed, slot returned; the producer stages the second,
    time.sleep(0.5)                                    # then waits for BOTH slots for the big one: one is held by 'ready'
    th, ref = feeder._th, weakref.ref(feeder)
    assert th.is_alive()
    del feeder
    gc.collect()
    th.join(timeout=10)
    assert ref() is None and not th.is_alive()


def test_host_feeder_passes_ready_batches_through_and_closes(hip):
    """harness.evaluate wraps every batch stream in a HostFeeder when a pipeline is given.  Batches that carry the
    reference's img / valid_mask (no 'events' entry, test.py:60) pass through unchanged, in order, between batches
    that are staged; a consumer that leaves the loop early (or an exception mid-iteration) must not leave the
    producer thread blocked on its queue holding the pinned and device rings; a single tensor where a list of
    samples is expected is refused instead of being iterated row by row."""
    import threading
    import time
    import torch
    from eventclip_amd.event2img import Event2ImagePipeline
    from eventclip_amd.synthetic import GEOMETRY, make_events
    g = GEOMETRY['n_cars']
    qa = dict(max_imgs=2, N=g['N'], split_method='event_count', convert_method='event_histogram', grayscale=True,
              count_non_zero=True, background_mask=False)
    pipe = Event2ImagePipeline(g['resolution'], g['max_n'], qa, n_px=224, patch=32, kpad=6144)
    ev = [make_events(12500, g['resolution'], seed=i) for i in range(3)]
    ready = dict(img=torch.zeros(2, 2, 3, 8, 8), valid_mask=torch.ones(2, 2, dtype=torch.bool), label=torch.tensor([0, 1]))
    stream = [dict(events=ev, label=torch.tensor([0, 1, 0])), ready, dict(events=ev[:1], label=torch.tensor([1]))]
    out = list(pipe.stream(iter(stream)))
    assert len(out) == 3 and out[1] is ready and 'patches' in out[0] and 'patches' in out[2]
    assert out[0]['label'].tolist() == [0, 1, 0] and out[2]['label'].tolist() == [1]
    # leaving early: close() ends the producer although it is ahead of the consumer
    before = threading.active_count()
    feeder = pipe.stream(iter([ev] * 50), depth=2)
    first = next(feeder)
    assert 'patches' in first
    feeder.close()
    feeder.close()                                   # idempotent
    deadline = time.time() + 10
    while feeder._th.is_alive() and time.time() < deadline:
        time.sleep(0.05)
    assert not feeder._th.is_alive() and threading.active_count() <= before + 1
    with pytest.raises(StopIteration):
        next(feeder)
    # context manager + an exception in the loop body
    with pytest.raises(KeyError):
        with pipe.stream(iter([ev] * 50)) as f2:
            for _ in f2:
                raise KeyError('consumer failed')
    assert not f2._th.is_alive() or (f2._th.join(5) or not f2._th.is_alive())
    # a single tensor is not a batch
    with pytest.raises(TypeError):
        list(pipe.stream(iter([torch.from_numpy(ev[0])])))
    # a feeder that is simply dropped: the producer holds only a weak reference, so the object is collected, closed by
    # its finaliser, and the thread ends (it kept the feeder and its pinned / device rings alive for ever before)
    import gc
    import weakref
    f3 = pipe.stream(iter([ev] * 50), depth=2)
    next(f3)
    th, ref = f3._th, weakref.ref(f3)
    del f3
    deadline = time.time() + 10
    while (ref() is not None or th.is_alive()) and time.time() < deadline:
        gc.collect()
        time.sleep(0.05)
    assert ref() is None and not th.is_alive()
