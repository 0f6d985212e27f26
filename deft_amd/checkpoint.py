"""Checkpoint loader for DEFT state dicts -- the tolerance rules of the reference's `load_model` (model/model.py:40-90), without a
torch module on the receiving side: the "created model" those rules compare against is the parameter table of the architecture
(deft_amd.synth._param_table: the reference's key names and shapes, validated against the reference's own DLASeg by oracle/make_golden.py).

    state_dict = load_model_state(path_or_checkpoint, opt)          # model.py:40-90
    model = deft_amd.integrate.create_model(opt, state_dict)

Rules mirrored (model.py line numbers):
  * `module.` prefixes of DataParallel checkpoints are stripped, `module_list...` keys are not (:49-53);
  * a loaded parameter whose shape differs from the model's -- or, with `opt.reset_hm`, any `hm*` parameter with 80 or 1 rows -- is
    either re-used row-wise (`opt.reuse_hm`: the first rows of the loaded tensor, :63-76 -- the reference's `<` test compares a
    shape with itself, so it always takes the slicing branch; a loaded tensor with FEWER rows than the model then fails inside
    `load_state_dict`, and does so here too) or skipped: the model keeps its freshly initialised parameter (:77-85);
  * loaded parameters the model does not have are dropped (:86-87); parameters the checkpoint lacks keep their initial value (:88-91).
"Freshly initialised" follows the reference's constructors where they are deterministic: BatchNorm (1, 0, 0, 1), the heat-map
head's last bias = `opt.prior_bias` (base_model.py:91-92, 99-100), the other heads' biases 0 (`fill_fc_weights`, :16-20), the
bilinear `up_*` weights (dla.py:565-573), DCNv2's zero-initialised `conv_offset_mask`.  Convolution weights the reference would
leave at their RANDOM initial value are zero here (a detector with such a layer is untrained either way); every such case is logged.
"""
import torch

from . import synth


def _strip_module(state_dict_):
    out = {}
    for k, v in state_dict_.items():
        out[k[7:] if k.startswith("module") and not k.startswith("module_list") else k] = v          # model.py:49-53
    return out


def model_template(opt=None, dataset=None, heads=None):
    """name -> freshly initialised tensor of the architecture `create_model(opt.arch = dla_34, opt.heads, opt.head_conv)` builds."""
    dataset = dataset or getattr(opt, "dataset", "mot")
    # this package builds ONE architecture: dla_34 with 256-wide one-layer head convs (what every DEFT experiment script trains:
    # experiments/*.sh --arch dla_34, opts.py:225-233 head_conv default for dla = 256).  Anything else must fail here, not load as an
    # all-initial template with every checkpoint key "dropped".
    arch = getattr(opt, "arch", "dla_34")
    if arch not in ("dla_34", "dla34"):
        raise ValueError("deft_amd builds the dla_34 detector only; opt.arch = %r" % (arch,))
    hc = getattr(opt, "head_conv", 256)
    if isinstance(hc, dict):                                           # opts.py:386-389: {head: [widths]}
        widths = {tuple(v) if isinstance(v, (list, tuple)) else (v,) for v in hc.values()}
        ok = widths <= {(256,)}
    else:
        ok = hc in (256, -1, None) and getattr(opt, "num_head_conv", 1) == 1
    if not ok:
        raise ValueError("deft_amd builds heads with one 256-wide 3x3 conv; opt.head_conv = %r, num_head_conv = %r"
                         % (hc, getattr(opt, "num_head_conv", 1)))
    if heads is None:
        heads = getattr(opt, "heads", None)
    table_ds = dataset if dataset in synth.HEADS else "mot"
    prior_bias = float(getattr(opt, "prior_bias", -4.6))
    sd = {}
    for name, shape, kind in synth._param_table(table_ds, dict(heads) if heads else None):
        if kind == "bn_w" or kind == "bn_v":
            t = torch.ones(shape)
        elif kind == "bn_n":
            t = torch.tensor(0, dtype=torch.long)
        elif kind == "up":
            t = synth._up_weight(shape)
        else:
            t = torch.zeros(shape)
        sd[name] = t
    for h in (heads or synth.HEADS[table_ds]):
        if "hm" in h:                                                  # base_model.py:91-92
            sd[h + ".2.bias"] = torch.full_like(sd[h + ".2.bias"], prior_bias)
    return sd


def load_model_state(source, opt=None, template=None, log=print):
    """source: a checkpoint path, a checkpoint dict ({"state_dict": ...}) or a bare state dict.  Returns the state dict the
    reference's `load_model(model, path, opt)` would leave in `model` (see the module docstring for the one difference)."""
    ck = torch.load(source, map_location="cpu") if isinstance(source, (str, bytes)) or hasattr(source, "read") else source
    if isinstance(ck, dict) and "state_dict" in ck:
        if "epoch" in ck:
            log("loaded {}, epoch {}".format(source if isinstance(source, str) else "<checkpoint>", ck["epoch"]))
        ck = ck["state_dict"]
    state_dict = _strip_module(ck)
    model_state_dict = template if template is not None else model_template(opt)
    reset_hm, reuse_hm = bool(getattr(opt, "reset_hm", False)), bool(getattr(opt, "reuse_hm", False))
    out = {}
    for k, v in state_dict.items():
        if k not in model_state_dict:
            log("Drop parameter {}.".format(k))                          # model.py:86-87
            continue
        want = model_state_dict[k]
        if tuple(v.shape) != tuple(want.shape) or (reset_hm and k.startswith("hm") and v.dim() > 0 and v.shape[0] in (80, 1)):
            if reuse_hm:
                log("Reusing parameter {}, required shape{}, loaded shape{}.".format(k, tuple(want.shape), tuple(v.shape)))
                piece = v[: want.shape[0]] if v.dim() > 0 else v         # model.py:71-75 (the branch the reference always takes)
                if tuple(piece.shape) != tuple(want.shape):
                    raise RuntimeError("size mismatch for {}: copying a param with shape {} from checkpoint, the shape in current "
                                       "model is {}.".format(k, tuple(piece.shape), tuple(want.shape)))        # load_state_dict's error
                out[k] = piece.clone()
            else:
                log("Skip loading parameter {}, required shape{}, loaded shape{}.".format(k, tuple(want.shape), tuple(v.shape)))
                out[k] = want                                            # model.py:77-85
        else:
            out[k] = v
    dropped = sum(1 for k in state_dict if k not in model_state_dict)
    zero_convs = []
    for k, want in model_state_dict.items():
        if k not in out:
            log("No param {}.".format(k))                                # model.py:88-91
            out[k] = want
            if k.endswith(".weight") and want.dim() == 4 and not bool(want.any()) and "conv_offset_mask" not in k:
                zero_convs.append(k)
    # loud, not a log line among hundreds: a checkpoint of another architecture, or one that lacks whole layers
    if state_dict and dropped > 0.5 * len(state_dict):
        raise ValueError("load_model_state: %d of %d checkpoint parameters are not parameters of the dla_34 detector this package builds "
                         "(first: %s) -- wrong architecture or key naming" % (dropped, len(state_dict), next(k for k in state_dict if k not in model_state_dict)))
    if zero_convs:
        import warnings
        warnings.warn("load_model_state: %d convolution weights are missing from the checkpoint and were left at ZERO (the reference would "
                      "leave them at their random initial value): %s%s" % (len(zero_convs), ", ".join(zero_convs[:4]), " ..." if len(zero_convs) > 4 else ""))
    return out
