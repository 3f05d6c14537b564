"""CPU oracle of the DeepCORO_CLIP contrastive training path — TEST INFRASTRUCTURE ONLY.

Plain-PyTorch (CPU, fp32) restatements of the reference's algorithms, each citing the reference
file:line it follows.  Pinned against golden vectors generated from the importable reference modules
(tests/golden/make_golden.py); the MViT backbone is "parity unpinned" (torchvision absent, see mvit.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(`deepcoro_clip_amd`) never does and fails loudly without its HIP library.
"""
