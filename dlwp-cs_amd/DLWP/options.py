"""
Engine options: the few behaviours of the engine that can be switched off, for A/B measurements and for the tests that pin both
settings to the same numbers.  ONE environment variable carries them: DLWPCS_OPTIONS="name=0,name=1,..." (read when an option
is asked for, so a test can set it around the construction of a model).  Everything not listed here is not configurable.

  graphs          training steps / rollouts with static shapes are captured into hipGraphs and replayed      (Model.use_graphs)
  premask         gradients travel pre-masked: act'(y) is applied where a gradient is produced                (Model, bf16 plans)
  wgrad_batch     the weight gradients of all layers of a step as ONE persistent launch                      (Model.batch_wgrad)
  fuse_adam       the optimizer inside the reduction of the batched weight gradients                          (Model.fuse_adam)
  fuse_pack       ... which also refreshes the packed bf16 operands: replayed steps start without packing     (Model.fuse_pack)
  fuse_head       output layer + loss + loss gradient + the layer's data gradient as one launch               (Model.fuse_head_loss)
  fold_loss_tail  the loss's last reduction stage inside the step's last launch                               (Model.fold_loss_tail)
  fuse_pool       2x2 average pooling written by the epilogue of the convolution in front of it               (Model.fuse_pool)
  fold_ring       (padded-grid data gradient) a pooled tensor's ring fix-up inside the pooling adjoint        (Model.fold_ring)
  cf_model        channels_first models run channels_last inside, one transpose per input / output            (Model)
  padded_io       bf16 rollouts with C % 8 != 0 keep their state padded to C + pad channels between passes    (Model)
  host_staging    fit() / predict() on host arrays stage batches in pinned memory on a copy stream            (Model)
  resident_data   fit() over several epochs keeps the arrays it uploaded during the first one in HBM            (Model)
  fold_head       inference: the pointwise output layer inside the epilogue of the convolution in front of it  (Model)
  dgrad_gather    bf16 data gradients in gather form (no halo ring, no fix-up launches); 0: padded grid        (ops)
  check_finite    fit() raises on a non-finite loss                                                            (Model.check_finite)
"""
import os

DEFAULTS = {
    'graphs': True, 'premask': True, 'wgrad_batch': True, 'fuse_adam': True, 'fuse_pack': True, 'fuse_head': True,
    'fold_loss_tail': True, 'fuse_pool': True, 'fold_ring': True, 'cf_model': True, 'padded_io': True, 'host_staging': True, 'resident_data': True,
    'dgrad_gather': True, 'fold_head': True, 'check_finite': False,
}


_parsed = {}


def _parse(env):
    """{name: bool} of one DLWPCS_OPTIONS string (cached per distinct string: option() sits on eager hot paths)."""
    hit = _parsed.get(env)
    if hit is None:
        hit = {}
        for item in env.split(','):
            item = item.strip()
            if not item:
                continue
            k, _, v = item.partition('=')
            k = k.strip()
            if k not in DEFAULTS:
                raise KeyError('DLWPCS_OPTIONS: unknown engine option %r (known: %s)' % (k, ', '.join(sorted(DEFAULTS))))
            hit[k] = v.strip() not in ('0', 'false', 'False', 'off', '')
        if len(_parsed) > 64:
            _parsed.clear()
        _parsed[env] = hit
    return hit


def option(name):
    """Current value of engine option `name` (DLWPCS_OPTIONS overrides the default)."""
    if name not in DEFAULTS:
        raise KeyError('unknown engine option %r (known: %s)' % (name, ', '.join(sorted(DEFAULTS))))
    return _parse(os.environ.get('DLWPCS_OPTIONS', '')).get(name, DEFAULTS[name])


def snapshot():
    """Every option's current value (a Model keeps one from its construction: its captured graphs belong to that configuration)."""
    over = _parse(os.environ.get('DLWPCS_OPTIONS', ''))
    return {k: over.get(k, v) for k, v in DEFAULTS.items()}
