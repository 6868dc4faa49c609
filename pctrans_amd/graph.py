"""HIP-graph replay of the forward-only head for small batches.

At batch 1-2 a head forward is ~1300 kernel launches and host-bound (12.5 ms eager at 512x512 on MI355X); captured
once in a HIP graph it replays in 7.6 ms (same results up to the run-to-run bf16 rounding of two library calls, see
DESIGN.md 4.8).  Every kernel of the package is launched on the
caller's current stream with no host synchronisation, so the whole forward is capturable; the only requirement is fixed
input shapes (one graph per shape).  At the benchmark batch (64) the step is GPU-bound and a graph changes nothing.

    fwd = GraphedForward(head, example_features, autocast_dtype=torch.bfloat16)
    predictions, mask_features = fwd(features)        # tensors are static buffers, overwritten by the next call
"""
import torch


class GraphedForward:
    def __init__(self, module, example_inputs, autocast_dtype=None, warmup=3):
        assert all(t.is_cuda for t in example_inputs.values()), "HIP graphs need device tensors"
        self.module = module
        self.autocast_dtype = autocast_dtype
        self.static_in = {k: v.clone() for k, v in example_inputs.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream (allocator, caches, checks)
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self._run()

    def _run(self):
        with torch.no_grad(), torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16,
                                            enabled=self.autocast_dtype is not None):
            return self.module(self.static_in)

    def __call__(self, inputs):
        for k, buf in self.static_in.items():
            src = inputs[k]
            if src.shape != buf.shape or src.dtype != buf.dtype:
                raise ValueError("GraphedForward was captured for %s %s, got %s %s" % (
                    tuple(buf.shape), buf.dtype, tuple(src.shape), src.dtype))
            buf.copy_(src)
        self.graph.replay()
        return self.static_out


class _BackboneTuple(torch.nn.Module):
    """backbone(volume) -> the feature maps as a tuple in a fixed key order (graphed callables return tensors / tuples).
    Holds the backbone as a sub-module so that its parameters are part of the captured backward."""

    def __init__(self, backbone, keys):
        super().__init__()
        self.backbone, self.keys = backbone, keys

    def forward(self, volume):
        f = self.backbone(volume)
        return tuple(f[k] for k in self.keys)


class _PixelDecoderTuple(torch.nn.Module):
    def __init__(self, pixel_decoder, keys):
        super().__init__()
        self.pixel_decoder, self.keys = pixel_decoder, keys

    def forward(self, *feats):
        mask_features, enc, multi = self.pixel_decoder.forward_features(dict(zip(self.keys, feats)))
        return (mask_features, enc) + tuple(multi)


def graph_training_front(model, example_volume, warmup=3):
    """Capture the STATIC-SHAPE front of a training step -- backbone and pixel decoder, forward AND backward -- as HIP graphs
    (torch.cuda.make_graphed_callables), in place.  At the reference's per-GPU batch (2 crops) a training step is ~7 500
    launches and host-bound (connectomics/engine/trainer.py:113-160 calls the model once per iteration); these two modules
    are about a third of the launches and their tensor shapes depend on the crop size only.  The transformer decoder and
    the criterion stay eager: their shapes follow the matching (lengths of the matched index lists).

    SINGLE-RANK or FROZEN-NORM models only.  The captured graphs replay the kernels that ran at capture time: a plain
    BatchNorm inside them keeps computing PER-RANK statistics even if the module is converted to SyncBatchNorm afterwards
    (parallel.make_parallel does that by default, as build.py:80-81), and a SyncBatchNorm cannot be captured at all (its
    all_gather would be recorded into the graph).  So: SyncBatchNorm inside the front raises here; training-mode BatchNorm
    inside the front raises when a process group with more than one rank is initialised; and parallel.make_parallel refuses
    to convert the norms of a model whose front is captured.  The shipped configurations freeze the backbone's norms
    (FrozenBN) and use GroupNorm in the pixel decoder's encoder levels, so the four-level front has no BatchNorm at all; the
    three-level FPN stage (`NORM: SyncBN` in the yamls) does, and is only capturable on one rank.

    Call once after the model is on the device and BEFORE wrapping it in DistributedDataParallel, outside autocast (the capture
    runs the two modules as the reference does, pixel decoder in fp32; a caller's autocast context would not reach into the
    replay, so the graphed wrappers refuse to run under one); training inputs must keep `example_volume`'s shape and dtype
    (another shape raises).  A captured model cannot be deep-copied or pickled as a whole (its forward closures would drive the
    ORIGINAL model's graphs): copy / save the state dict, or capture again.  The modules stay registered where they are (parameters,
    state dict, `.eval()` paths are untouched: only their training-mode forward replays the graphs); BatchNorm statistics
    moved by the capture's warm-up passes are put back.  Measured (tools/record_train_configs.py --graph-front): configs[2]
    102.4 -> 94.3 ms per step, configs[3] 129.1 -> 120.1 ms.  Returns the model."""
    head = model.sem_seg_head
    backbone, pixel_decoder = model.backbone, head.pixel_decoder
    if "_pct_graphed" in backbone.__dict__:
        raise RuntimeError("graph_training_front: this model's front is already captured")
    if "_pct_graphed_core" in head.predictor.__dict__:
        raise RuntimeError("graph_training_front: this model's decoder core is captured (graph_training_decoder) -- use one "
                           "capture or the other (see graph_training_decoder)")
    import torch.distributed as dist
    multi_rank = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    for owner, mod in (("backbone", backbone), ("pixel_decoder", pixel_decoder)):
        for name, m in mod.named_modules():
            if isinstance(m, torch.nn.SyncBatchNorm):
                raise RuntimeError("graph_training_front: %s.%s is a SyncBatchNorm -- its cross-rank all_gather cannot be "
                                   "captured into a HIP graph; capture the front on a single-rank / frozen-norm model only"
                                   % (owner, name))
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and multi_rank:
                raise RuntimeError("graph_training_front: %s.%s is a BatchNorm and %d ranks are running -- the captured graphs "
                                   "would keep per-rank statistics where the reference synchronises them (build.py:80-81); "
                                   "the graphed front is single-rank or frozen-norm only" % (owner, name, dist.get_world_size()))
    if torch.is_autocast_enabled():
        raise RuntimeError("graph_training_front: call outside torch.autocast (see the docstring)")
    was_training = model.training
    model.train()
    owned = [("backbone." + n, b) for n, b in backbone.named_buffers()] + \
            [("pixel_decoder." + n, b) for n, b in pixel_decoder.named_buffers()]
    buffers = {n: b.detach().clone() for n, b in owned}
    with torch.no_grad():
        feats = backbone(example_volume)
    keys = sorted(feats)
    bb, pd = _BackboneTuple(backbone, keys), _PixelDecoderTuple(pixel_decoder, keys)
    sample_feats = tuple(feats[k].detach().clone().requires_grad_(True) for k in keys)
    g_bb, g_pd = torch.cuda.make_graphed_callables((bb, pd), ((example_volume.detach().clone(),), sample_feats),
                                                   num_warmup_iters=warmup)
    with torch.no_grad():                                 # the warm-up / capture passes ran BatchNorm in training mode
        for n, b in owned:                                # (keys carry the owner's prefix: the two modules may share names)
            b.copy_(buffers[n])
    eager_backbone, eager_features = backbone.forward, pixel_decoder.forward_features
    shape, dtype = tuple(example_volume.shape), example_volume.dtype

    def _no_autocast():
        if torch.is_autocast_enabled():
            raise RuntimeError("the graphed training front was captured outside autocast and replays as captured: run it "
                               "outside torch.autocast (the eager path would follow the caller's autocast, the replay cannot)")

    def _no_copy(*_a, **_k):
        raise RuntimeError("a model whose training front is captured (graph_training_front) cannot be deep-copied or pickled: "
                           "its forward closures drive the original model's HIP graphs -- copy the state dict instead")

    def backbone_forward(volume):
        if not (backbone.training and torch.is_grad_enabled()):
            return eager_backbone(volume)
        _no_autocast()
        if tuple(volume.shape) != shape or volume.dtype != dtype:
            raise ValueError("graph_training_front was captured for %s %s, got %s %s" % (shape, dtype, tuple(volume.shape),
                                                                                        volume.dtype))
        return dict(zip(keys, g_bb(volume)))

    def forward_features(features):
        if not (pixel_decoder.training and torch.is_grad_enabled()):
            return eager_features(features)
        _no_autocast()
        out = g_pd(*[features[k] for k in keys])
        return out[0], out[1], list(out[2:])

    backbone.__dict__["_pct_graphed"] = (g_bb, eager_backbone)
    pixel_decoder.__dict__["_pct_graphed"] = (g_pd, eager_features)
    backbone.forward = backbone_forward                   # instance attributes: the modules stay registered as they are
    pixel_decoder.forward_features = forward_features
    for m in (backbone, pixel_decoder):
        m.__dict__["__deepcopy__"] = _no_copy             # copy.deepcopy looks this up on the instance
        m.__dict__["__reduce_ex__"] = _no_copy            # pickle / torch.save(model) of the whole module
    model.train(was_training)
    return model


class _DecoderCore(torch.nn.Module):
    """The transformer decoder's static-shape tensor flow as a module of its own (tensors in, a flat tuple of tensors out);
    holds the decoder as a sub-module so that its parameters are part of the captured backward."""

    def __init__(self, decoder):
        super().__init__()
        self.decoder = decoder

    def forward(self, mask_features, *multi_scale_features):
        return self.decoder._forward_core(mask_features, *multi_scale_features)


def has_graphed_decoder(model):
    return "_pct_graphed_core" in model.sem_seg_head.predictor.__dict__


def graph_training_decoder(model, example_volume, warmup=3):
    """Capture the transformer decoder's STATIC-SHAPE core -- input projections, the nine cross-attention / self-attention /
    FFN layers, reference-point updates and the ten dynamic mask heads, forward AND backward -- as HIP graphs
    (torch.cuda.make_graphed_callables), in place.  This is the host-bound part of a training step at the reference's
    per-GPU batch of two crops (thousands of launches on [Q, N, C]-sized tensors); what follows the matching -- the ten
    Hungarian assignments, the query-contrast items and the criterion -- and the semantic head stay eager.
    `MultiScaleMaskedTransformerDecoder.forward` evaluates the core first and the matching afterwards in the eager path too,
    so eager and replayed steps run the same operations in the same order.

    The core holds no BatchNorm (the decoder's only one, SyncBN in the shipped yamls, sits in the semantic head, which is
    outside), no dropout (checked: a captured mask would repeat) and no collective: it can be captured on a model whose norms
    are already SyncBatchNorm and on every rank of a DistributedDataParallel job -- call it after parallel.convert_norms and
    BEFORE wrapping the model in DDP (then make_parallel(..., norm_mode=None)); the gradients leave the graph as ordinary
    autograd outputs, so DDP's bucket hooks fire as in the eager step (tools/rehearse_ddp_sync_bn.py --graph-decoder, two
    ranks: gradients bitwise equal across ranks).  Outside torch.autocast; fixed crop size and batch; no deep copy / pickle of
    the captured model.  NOT combinable with graph_training_front (either raises when the other is in place): measured, the
    decoder core alone is the faster capture (configs[2]: 72.2 ms against 81.2 for the front), and with both in place a
    backward graph launch crashed the HIP runtime in two of four processes of tools/record_train_configs.py.  Returns the model."""
    head = model.sem_seg_head
    decoder = head.predictor
    if "_pct_graphed_core" in decoder.__dict__:
        raise RuntimeError("graph_training_decoder: this model's decoder is already captured")
    if "_pct_graphed" in model.backbone.__dict__ or "_pct_graphed" in head.pixel_decoder.__dict__:
        raise RuntimeError("graph_training_decoder: this model's front is captured (graph_training_front) -- use one capture or "
                           "the other: the decoder core alone is the faster of the two, and with both captured the HIP "
                           "runtime crashed inside a backward graph launch in two of four processes (DESIGN.md 4.7)")
    for name, m in decoder.named_modules():
        if name == "logits" or name == "seg_head" or name.startswith("seg_head."):
            continue                                      # the semantic head runs outside the core
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            raise RuntimeError("graph_training_decoder: predictor.%s is a BatchNorm inside the decoder core -- its statistics "
                               "(and, as SyncBatchNorm, its all_gather) cannot be part of a captured graph" % name)
        if isinstance(m, torch.nn.Dropout) and m.p > 0:
            raise RuntimeError("graph_training_decoder: predictor.%s has dropout %g -- a captured mask would repeat" % (name, m.p))
        if isinstance(getattr(m, "dropout", None), float) and m.dropout > 0:
            raise RuntimeError("graph_training_decoder: predictor.%s has attention dropout %g" % (name, m.dropout))
    if torch.is_autocast_enabled():
        raise RuntimeError("graph_training_decoder: call outside torch.autocast (see graph_training_front)")
    was_training = model.training
    # example inputs of the core (only their shapes are used): one pass through backbone + pixel decoder in EVAL mode -- no
    # BatchNorm statistics move and no SyncBatchNorm exchange runs
    model.eval()
    with torch.no_grad():
        feats = model.backbone(example_volume)
        mask_features, _enc, multi = head.pixel_decoder.forward_features(feats)
    model.train()
    sample = tuple(t.detach().clone().requires_grad_(True) for t in (mask_features,) + tuple(multi))
    shapes = tuple((tuple(t.shape), t.dtype) for t in sample)
    g_core = torch.cuda.make_graphed_callables(_DecoderCore(decoder), sample, num_warmup_iters=warmup,
                                               allow_unused_input=True)

    def core(mask_features, *multi_scale_features):
        if torch.is_autocast_enabled():
            raise RuntimeError("the graphed decoder core was captured outside autocast and replays as captured: run it "
                               "outside torch.autocast")
        got = tuple((tuple(t.shape), t.dtype) for t in (mask_features,) + tuple(multi_scale_features))
        if got != shapes:
            raise ValueError("graph_training_decoder was captured for %s, got %s" % (shapes, got))
        return g_core(mask_features, *multi_scale_features)

    def _no_copy(*_a, **_k):
        raise RuntimeError("a model whose decoder is captured (graph_training_decoder) cannot be deep-copied or pickled: "
                           "its forward drives the original model's HIP graphs -- copy the state dict instead")

    decoder.__dict__["_pct_graphed_core"] = core
    decoder.__dict__["__deepcopy__"] = _no_copy
    decoder.__dict__["__reduce_ex__"] = _no_copy
    model.train(was_training)
    return model


def release_training_graphs(model):
    """Undo graph_training_front / graph_training_decoder: the modules run eagerly again and the captured HIP graphs (with their
    private memory pools) are destroyed HERE, on the calling thread, after the device has drained -- not whenever Python's
    cyclic collector finds the closures (which may be inside another model's replay, on autograd's worker thread: destroying a
    graph while one is being launched crashed the HIP runtime in tools/record_train_configs.py, which builds several captured
    models in one process).  Call it before dropping a captured model in a process that goes on using the device."""
    import gc
    head = model.sem_seg_head
    torch.cuda.synchronize()
    for mod, attr in ((model.backbone, "forward"), (head.pixel_decoder, "forward_features")):
        if "_pct_graphed" in mod.__dict__:
            del mod.__dict__["_pct_graphed"]
            mod.__dict__.pop(attr, None)                  # the instance attribute shadowing the class's eager method
            mod.__dict__.pop("__deepcopy__", None)
            mod.__dict__.pop("__reduce_ex__", None)
    dec = head.predictor
    if "_pct_graphed_core" in dec.__dict__:
        del dec.__dict__["_pct_graphed_core"]
        dec.__dict__.pop("__deepcopy__", None)
        dec.__dict__.pop("__reduce_ex__", None)
    gc.collect()
    torch.cuda.synchronize()
    return model
