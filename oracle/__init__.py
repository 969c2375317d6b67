"""CPU oracle for the EventCLIP hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, what the reference computes on the path
events -> histogram frames -> CLIP preprocess -> CLIP ViT / text encoder ->
adapter -> logits.  It exists so that the HIP path in ``eventclip_amd`` can be
checked against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never
does, and fails loudly when its HIP library is missing.

Pinning (what anchors each restatement):

* ``oracle.events``   -- pinned: bit-exact against the reference's own
  ``datasets/vis.py`` imported in the build container; fixtures in
  ``tests/golden/events_*.npz`` (generator: ``tools/make_golden_events.py``).
* ``oracle.classify`` / ``oracle.adapter`` -- pinned against the reference's
  ``models/clip_cls.py`` / ``models/adapter.py`` imported in the build
  container (``tools/make_golden_models.py``).
* ``oracle.event_utils`` -- pinned: ``center_events`` / flips against the reference's
  ``datasets/utils.py`` (``tools/make_golden_event_utils.py``) and the N-ImageNet reader against
  the reference's ``datasets/imagenet.py:load_event`` (``tools/make_golden_ingest.py``).
* ``oracle.pseudo_label`` -- pinned: ``gen_data.py``'s ``main()`` itself is run with stand-ins for clip /
  nerv / models / datasets around the selection code (lines 132-164, 196-215) and the pseudo-labels it
  writes are the fixture (``tools/make_golden_pseudo.py``).
* ``oracle.train`` -- pinned: loss and gradients of the `text-identity` and `text-trans` few-shot steps
  against the reference's own ``FSCLIPClassifier`` under torch autograd
  (``tools/make_golden_train.py``); Adam against ``torch.optim.Adam``; the warm-up + cosine schedule is
  PARITY UNPINNED (it lives in the absent ``nerv``).
* ``oracle.preprocess`` -- the reference calls un-vendored ``clip._transform``
  (torchvision Resize/CenterCrop/ToTensor/Normalize over PIL).  Pinned against
  PIL itself (``Image.resize(BICUBIC)``), which is what torchvision calls.
* ``oracle.clip_ref`` -- PARITY UNPINNED against the reference: the CLIP
  ViT / text arithmetic lives in un-vendored ``openai/CLIP`` (``clip==1.0``,
  git HEAD) and no weights or golden vectors exist upstream.  The restatement
  follows the published architecture and is cross-checked against HF
  ``transformers`` CLIP with seeded random weights: a tiny model for both towers
  (``tools/make_golden_clip.py``) and the vision tower at the full ViT-L/14 / ViT-B/32 geometry and
  depth (``tools/make_golden_vit.py``).
"""
