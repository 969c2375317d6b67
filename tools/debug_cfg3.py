"""configs[3] (signal weights) through a few tower modes: error vs the fp32 oracle (diagnostic; round 5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
# EC_ATTN_SPLIT_F32 is read by the DIAGNOSTIC build only since round 6 (the product library's kernel choice never depends on the environment)
os.environ.setdefault('EVENTCLIP_HIP_LIB', os.path.join(ROOT, 'eventclip_amd', 'libeventclip_hip_diag.so'))
import torch
import test_configs_gpu as tc
from eventclip_amd import clip as eclip
from eventclip_amd.clip_cls import ZSCLIPClassifier
from eventclip_amd.event2img import Event2ImagePipeline
g, qa = tc.quantize_args('n_imagenet', 2)
cfg = eclip.arch_config('ViT-L/14@336px', text_layers=1)
sd = tc.make_weights('n_imagenet/ViT-L/14@336px', cfg, 33, 'signal')
tokens = eclip.synthetic_tokens(1000, seed=3)
evs = tc.make_events_batch(2, [135000, 70000], g['resolution'], 3, 'signal')
want, feats = tc.oracle_forward(evs, g['resolution'], qa, cfg, sd, tokens, 2, 'mean')
for name, kw, env in (('default', {}, {}), ('precise (all blocks)', dict(image_precise=True), {}),
                      ('8:5 hl2 attention', dict(image_precise_blocks=8), {}),
                      ('8:5 fp32 attention kernel', dict(image_precise_blocks=8), {'EC_ATTN_SPLIT_F32': '1'}),
                      ('8:8 hl2', dict(image_precise_blocks=8, image_precise_attn_blocks=8), {}),
                      ('12:12 hl2', dict(image_precise_blocks=12, image_precise_attn_blocks=12), {}),
                      ('23:23 hl2', dict(image_precise_blocks=23, image_precise_attn_blocks=23), {})):
    for k in ('EC_ATTN_SPLIT_F32',):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = eclip.CLIP(cfg, sd, **kw).cuda().eval()
    model = ZSCLIPClassifier(clip_dict=dict(clip_model=m, prompt='a point cloud image of a {}', class_names=[str(i) for i in range(1000)],
                                            agg_func='mean', class_tokens=tokens)).cuda().eval()
    pipe = Event2ImagePipeline(g['resolution'], g['max_n'], qa, n_px=336, patch=14, kpad=m.kpad)
    out = model(pipe(evs))
    e = tc.logit_errors({k: v.cpu() for k, v in out.items()}, want)
    print(f'{name:28s}: full_logits {e["full_logits"][0]:.2e} logits {e["logits"][0]:.2e}', flush=True)
    del m, model
