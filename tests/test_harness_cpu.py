"""Host-side logic on CPU: chunking, view planning, meters, and the data-parallel
shard / all-gather path over gloo with world_size 2 (the oracle stands in for the GPU
forward here -- this tests the sharding harness, not the kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eventclip_amd import harness, vis


def test_chunk_bounds_match_oracle_split():
    from oracle import events as oe
    for tot in [1, 5, 10, 14, 15, 16, 20, 25, 26, 30, 100, 104, 105, 106, 20000, 225000]:
        for N in (10, 7, 20000):
            assert vis.chunk_bounds(tot, N) == oe.split_event_count(tot, N)


def test_split_event_count_signature():
    t = np.linspace(0, 1, 26)
    i0, i1, t0, t1 = vis.split_event_count(t, 10)
    assert i0 == [0, 10, 16] and i1 == [10, 20, 26]
    np.testing.assert_array_equal(t0, t[[0, 10, 16]])
    np.testing.assert_array_equal(t1, t[[9, 19, 25]])


def test_colour_map():
    from oracle import events as oe
    for g in (True, False, 200, [90, 127, 255]):
        a, b = vis.colour_map(g), oe.colour_map(g)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])


def test_pipeline_plan_padding_and_subsampling():
    from eventclip_amd.event2img import Event2ImagePipeline
    qa = dict(max_imgs=3, N=100, split_method='event_count', convert_method='event_histogram',
              grayscale=True, count_non_zero=False, background_mask=True)
    pipe = Event2ImagePipeline((36, 52), 1000, qa, generator=torch.Generator().manual_seed(0))
    assert pipe.max_imgs == 3                       # min(round(1000/100), 3), event2img.py:70-72
    fr, ri, vm = pipe.plan([250, 60, 1000])
    # sample 0: 2 chunks + dropped remainder; sample 1: one short chunk; sample 2: 10 chunks -> 3
    assert vm.tolist() == [[True, True, False], [True, False, False], [True, True, True]]
    assert fr[:3].tolist() == [[0, 100], [100, 200], [250, 310]]
    sel = fr[3:] - 310
    assert all((b - a) == 100 and a % 100 == 0 for a, b in sel.tolist())
    assert ri.tolist()[0] == [0, 1, -1] and ri.tolist()[1] == [2, -1, -1]
    with pytest.raises(IndexError):
        pipe.plan([0])
    qa2 = dict(qa, convert_method='voxel')
    with pytest.raises(NotImplementedError):
        Event2ImagePipeline((36, 52), 1000, qa2)
    # N-Cars: round(12500 / 30000) = 0 -> at least one view (event2img.py:72)
    assert Event2ImagePipeline((100, 120), 12500, dict(qa, N=30000)).max_imgs == 1


def test_average_meter_and_accuracies():
    m = harness.AverageMeter()
    m.update(1.0, 3)
    m.update(0.0, 1)
    assert abs(m.avg - 0.75) < 1e-12                # sum(acc*n)/sum(n), test.py:67
    probs = torch.tensor([[0.1, 0.9], [0.8, 0.2]])
    out = dict(probs=probs, logits=probs.flip(1))
    acc = harness.batch_accuracies(out, torch.tensor([1, 1]))
    assert acc == {'probs_acc': 0.5, 'logits_acc': 0.5}


def test_shard_range_partitions():
    for n in (0, 1, 7, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [harness.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import classify as oc
    g = torch.Generator().manual_seed(0)
    B, T, C, K = 7, 3, 16, 5                        # uneven shards on purpose
    valid = torch.rand(B, T, generator=g) < 0.7
    valid[:, 0] = True
    feats = torch.randn(B, T, C, generator=g)
    text = torch.nn.functional.normalize(torch.randn(K, C, generator=g), dim=-1)
    lo, hi = harness.shard_range(B, rank, world)
    local = oc.zs_forward(feats[lo:hi][valid[lo:hi]], valid[lo:hi], text, 100.0, 'mean')
    sizes = [harness.shard_range(B, r, world)[1] - harness.shard_range(B, r, world)[0]
             for r in range(world)]
    gathered = harness.gather_out_dict(local, sizes)
    full = oc.zs_forward(feats[valid], valid, text, 100.0, 'mean')
    ok = all(torch.allclose(gathered[k], full[k], atol=1e-5) for k in ('logits', 'probs'))
    # even shards take the single all_gather_into_tensor path
    even = harness.all_gather_rows(torch.full((2, 3), float(rank)))
    ok = ok and even.shape == (2 * world, 3) and even[2 * rank].eq(rank).all().item()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_forward_and_all_gather_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_sharded_forward_and_all_gather_gloo_world8():
    """Eight ranks (one node of the scaling run): B = 7 samples over 8 ranks leaves rank 7 with an EMPTY shard and
    the others with one sample each -- the all-gather's padding path with a zero-row rank."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(8)]


def test_product_fails_loudly_without_gpu():
    from eventclip_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.HipLibraryError):
        vis.events2frames(np.zeros((4, 4), np.float32), 'event_count', 'event_histogram', N=2)
    from eventclip_amd import clip as eclip
    cfg = eclip.arch_config('ViT-B/32', layers=1, text_layers=1, vocab_size=64)
    m = eclip.CLIP(cfg, eclip.random_state_dict(cfg, 0))
    with pytest.raises(_lib.HipLibraryError):
        m.encode_image(torch.zeros(1, 3, 224, 224))


def test_clip_surface_without_gpu():
    from eventclip_amd import clip as eclip
    assert 'ViT-L/14' in eclip.available_models()
    with pytest.raises(NotImplementedError):
        eclip.arch_config('RN50')
    with pytest.raises(RuntimeError):
        eclip.arch_config('ViT-Z/1')
    cfg = eclip.arch_config('ViT-B/32', layers=2, text_layers=1, vocab_size=64)
    sd = eclip.random_state_dict(cfg, 0)
    m = eclip.CLIP(cfg, sd)
    assert m.visual.output_dim == 512 and m.logit_scale.ndim == 0
    assert abs(m.logit_scale.exp().item() - 100.0) < 1e-3
    assert set(m.state_dict()) == set(sd)           # OpenAI's key names round-trip
    assert eclip.config_from_state_dict(sd)['layers'] == 2
    tok = eclip.synthetic_tokens(4, seed=1)
    assert tok.shape == (4, 77) and (tok.argmax(-1) >= 5).all()
    with pytest.raises(FileNotFoundError):
        eclip.load('ViT-B/32', download_root='/nonexistent')


def test_tokenize_with_synthetic_bpe_vocab(tmp_path, monkeypatch):
    """clip.tokenize (models/clip_cls.py:81-83): byte-level BPE, <SOT> ... <EOT>, zero padding,
    truncation rule.  The real vocabulary file is not shipped, so a tiny merges file stands in."""
    import gzip
    from eventclip_amd import clip as eclip
    merges = ['#version: test', 'c a', 'ca t</w>', 'd o', 'do g</w>', 'a</w> x</w>']
    path = tmp_path / 'bpe.txt.gz'
    with gzip.open(path, 'wt', encoding='utf-8') as f:
        f.write('\n'.join(merges) + '\n')
    monkeypatch.setenv('EVENTCLIP_BPE_PATH', str(path))
    eclip._tokenizer.cache_clear()
    tk = eclip._tokenizer()
    sot, eot = tk.encoder['<|startoftext|>'], tk.encoder['<|endoftext|>']
    assert eot == sot + 1 == len(tk.encoder) - 1
    out = eclip.tokenize(['a photo of a cat', 'DOG'])
    assert out.shape == (2, 77) and out.dtype == torch.int32
    assert out[0, 0] == sot and out[1, 0] == sot
    assert int(out[1].argmax()) == 2 and out[1, 2] == eot            # <SOT> dog</w> <EOT>
    assert out[1, 1] == tk.encoder['dog</w>'] and (out[1, 3:] == 0).all()
    assert tk.encoder['cat</w>'] in out[0].tolist()
    assert eclip.tokenize('cat').shape == (1, 77)                      # str input
    long = ' '.join(['cat'] * 100)
    with pytest.raises(RuntimeError):
        eclip.tokenize(long)
    t = eclip.tokenize(long, truncate=True)
    assert t[0, 76] == eot and t[0, 0] == sot
    eclip._tokenizer.cache_clear()
    monkeypatch.delenv('EVENTCLIP_BPE_PATH')
    with pytest.raises(FileNotFoundError):
        eclip.tokenize('cat')
    eclip._tokenizer.cache_clear()


def test_meters_match_the_references_test_py_main():
    """tests/golden/eval_meters.npz: accuracies printed / returned by the reference's own test.py main()
    (driven with stand-ins by tools/make_golden_eval.py) for uneven batches, top-1 and top-5."""
    import os
    import numpy as np
    import torch
    from conftest import GOLDEN
    from eventclip_amd.harness import AverageMeter, batch_accuracies
    z = np.load(os.path.join(GOLDEN, 'eval_meters.npz'))
    for name, top5 in (('ncaltech', False), ('nin', True)):
        labels, logits, probs = (torch.from_numpy(z[f'{name}_{k}']) for k in ('labels', 'logits', 'probs'))
        meters, i0 = {}, 0
        for n in z[name + '_sizes']:
            sl = slice(i0, i0 + int(n))
            acc = batch_accuracies({'probs': probs[sl], 'logits': logits[sl]}, labels[sl], top5=top5)
            for k, v in acc.items():
                meters.setdefault(k, AverageMeter()).update(v, int(n))
            i0 += int(n)
        got = {k: m.avg for k, m in meters.items()}
        assert abs(got['probs_acc'] - float(z[name + '_acc1'][0])) < 1e-12
        assert abs(got['logits_acc'] - float(z[name + '_acc1'][1])) < 1e-12
        printed = z[name + '_printed']
        order = ['probs_acc', 'logits_acc'] + (['probs_acc5', 'logits_acc5'] if top5 else [])
        assert len(printed) == len(order)
        for k, want in zip(order, printed):
            assert abs(round(got[k] * 100., 2) - float(want)) < 1e-9, (name, k)


def _e2i():
    import os
    import numpy as np
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'event2img.npz'))
    qa = dict(max_imgs=int(z['qa_max_imgs']), split_method=str(z['qa_split_method']),
              convert_method=str(z['qa_convert_method']), N=int(z['qa_N']), grayscale=bool(z['qa_grayscale']),
              count_non_zero=bool(z['qa_count_non_zero']), background_mask=bool(z['qa_background_mask']))
    return z, qa


def test_view_planning_matches_the_references_event2image_dataset():
    """max_imgs, valid masks and view counts of the reference's own Event2ImageDataset
    (tests/golden/event2img.npz, tools/make_golden_event2img.py) from Event2ImagePipeline.plan (host only)."""
    import torch
    from eventclip_amd.event2img import Event2ImagePipeline
    z, qa = _e2i()
    pipe = Event2ImagePipeline(tuple(int(v) for v in z['resolution']), int(z['max_n']), qa, n_px=224)
    assert pipe.max_imgs == int(z['max_imgs'])
    for i, n in enumerate(z['counts']):
        torch.manual_seed(1000 + i)
        fr, ri, vm = pipe.plan([int(n)])
        assert vm[0].tolist() == z[f'tta0_valid{i}'].tolist()
        assert fr.shape[0] == int(z[f'tta0_valid{i}'].sum())


def test_lora_fold_matches_the_references_effective_weights():
    """eventclip_amd.lora.merge_lora_visual against the reference's own LoRA modules (tests/golden/lora.npz,
    tools/make_golden_lora.py): int rank (q, k, v), 'qv-3' (no k factors) and 'qkvo-2' (out_proj too)."""
    import os
    import numpy as np
    import torch
    from conftest import GOLDEN
    from eventclip_amd.lora import load_finetuned_visual, merge_lora_visual
    z = np.load(os.path.join(GOLDEN, 'lora.npz'))
    for tag in ('r4', 'qv', 'qkvo'):
        sd = {k.split('sd:')[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + '/sd:')}
        base = {k.split('base:')[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + '/base:')}
        merged = merge_lora_visual(sd)
        assert sorted(merged) == sorted(base)                         # plain nn.MultiheadAttention keys again
        x = torch.from_numpy(z[tag + '/x'])
        for i in range(2):
            pre = f'transformer.resblocks.{i}.attn.'
            for name in ('in_proj_weight', 'out_proj.weight'):
                want = z[f'{tag}/eff:{pre}{name}']
                np.testing.assert_allclose(merged[pre + name].numpy(), want, rtol=1e-6, atol=1e-7)
            assert torch.equal(merged[pre + 'in_proj_bias'], base[pre + 'in_proj_bias'])
            # a plain MHA with the folded weights reproduces the LoRA module's output
            mha = torch.nn.MultiheadAttention(x.shape[-1], 2).eval()
            mha.load_state_dict({k[len(pre):]: v for k, v in merged.items() if k.startswith(pre)})
            with torch.no_grad():
                y = mha(x, x, x, need_weights=False)[0]
            np.testing.assert_allclose(y.numpy(), z[f'{tag}/y{i}'], rtol=1e-5, atol=1e-6)
        if tag == 'qv':       # no k factors: the k block is the frozen weight
            w = merged['transformer.resblocks.0.attn.in_proj_weight']
            assert torch.equal(w[16:32], sd['transformer.resblocks.0.attn.in_proj_weight.merged_proj'][16:32])
    ck = {'state_dict': {'model.visual.' + k: v for k, v in sd.items()} | {'text_feats': torch.zeros(2, 4)}}
    full = load_finetuned_visual({'visual.conv1.weight': torch.ones(1), 'token_embedding.weight': torch.ones(1)}, ck)
    assert 'token_embedding.weight' in full and 'visual.conv1.weight' not in full
    assert 'visual.transformer.resblocks.1.attn.out_proj.weight' in full and not any('lora' in k for k in full)


def test_custom_ops_are_registered_and_have_no_cpu_kernel():
    """north_star's boundary: `eventclip_hip::` PyTorch custom ops over the C ABI.  Registered for
    CUDA (HIP) tensors only: shapes come from the fake kernels, CPU tensors are refused."""
    import pytest
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from eventclip_amd import torch_ops
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.eventclip_hip, name), name
    with FakeTensorMode():
        f = torch.empty(12, 768, device='cuda')
        ri = torch.zeros(4, 3, dtype=torch.int32, device='cuda')
        full, logits, probs = torch.ops.eventclip_hip.classify(f, ri, torch.empty(768, 101, device='cuda'),
                                                               100.0, 1, False)
        assert full.shape == (4, 3, 101) and logits.shape == (4, 101) and probs.shape == (4, 101)
        ev = torch.empty(1000, 4, device='cuda')
        fr = torch.zeros(5, 2, dtype=torch.int64, device='cuda')
        frames = torch.ops.eventclip_hip.events_to_frames(ev, fr, 180, 240, 10., [255, 0, 0], [0, 0, 255], False,
                                                          True, 0, False, False, False, 0)
        assert frames.shape == (5, 180, 240, 3) and frames.dtype == torch.uint8
        pt = torch.ops.eventclip_hip.preprocess(frames, 224, 1, 14, 1216, 0)
        assert pt.shape == (5, 256, 1216) and pt.dtype == torch.float16
    with pytest.raises(NotImplementedError):
        torch.ops.eventclip_hip.classify(torch.zeros(2, 4), torch.zeros(1, 2, dtype=torch.int32),
                                         torch.zeros(4, 3), 1.0, 1, False)
