"""Checkpoints with the reference's file naming (audiogan.py:936-939: ``%s-{dis,gen,eg,ed}-%05d``)
but as ``state_dict``s instead of whole-module pickles, so they load into this package's modules AND
into the reference's (the parameter names are identical, see tests/test_host_logic.py).  Optimiser
state, step counters and RNG state -- which the reference never saved -- go
into a fifth file ``%s-opt-%05d``.

``load`` also reads the REFERENCE's own files: those hold whole-module pickles (``T.save(d, ...)``, audiogan.py:936-939,
read back with ``T.load`` at :698-701); their ``state_dict()`` is taken (unpickling them needs the reference's classes
importable, as it does for the reference itself).  The RNG streams (torch CPU + every CUDA device) saved in the
``opt`` file are restored, so a resumed run draws the same z / noise / stop decisions."""
import os

import torch

_ROLES = (('dis', 'd'), ('gen', 'g'), ('eg', 'e_g'), ('ed', 'e_d'))


def _path(prefix, role, iteration):
    return '%s-%s-%05d' % (prefix, role, iteration)


def _plain(x, where='extra'):
    """``extra`` must survive ``torch.load(weights_only=True)``: tensors and plain Python containers / scalars only (numpy
    scalars and 0-d arrays are converted; anything else is refused here, at save time, not at the next resume)"""
    import numbers
    if x is None or isinstance(x, (bool, int, float, str, bytes)) or torch.is_tensor(x):
        return x
    if isinstance(x, dict):
        return {_plain(k, where): _plain(v, '%s[%r]' % (where, k)) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v, where + '[...]') for v in x)
    if hasattr(x, 'dtype') and hasattr(x, 'shape'):          # numpy scalar / array
        import numpy as np
        a = np.asarray(x)
        return a.item() if a.ndim == 0 else torch.from_numpy(np.ascontiguousarray(a))
    if isinstance(x, numbers.Number):
        return x.real if isinstance(x, numbers.Real) else complex(x)
    raise TypeError('checkpoint.save: %s holds a %s; only tensors, numpy arrays and plain Python values can be stored '
                    '(the file must load with weights_only=True)' % (where, type(x).__name__))


def save(prefix, iteration, d=None, g=None, e_g=None, e_d=None, opt_d=None, opt_g=None, extra=None):
    extra = _plain(extra)
    mods = dict(d=d, g=g, e_g=e_g, e_d=e_d)
    written = []
    for role, key in _ROLES:
        m = mods[key]
        if m is not None:
            torch.save({k: v.detach().cpu() for k, v in m.state_dict().items()}, _path(prefix, role, iteration))
            written.append(_path(prefix, role, iteration))
    if opt_d is not None or opt_g is not None or extra is not None:
        blob = dict(opt_d=_cpu(opt_d.state_dict()) if opt_d is not None else None,
                    opt_g=_cpu(opt_g.state_dict()) if opt_g is not None else None,
                    extra=extra, torch_rng=torch.get_rng_state(),
                    cuda_rng=torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None)
        torch.save(blob, _path(prefix, 'opt', iteration))
        written.append(_path(prefix, 'opt', iteration))
    return written


def _read(path, allow_pickle):
    """state_dict files (what ``save`` writes) load with ``weights_only=True``.  The reference's whole-module pickles
    execute code when unpickled: they are read only when the caller says so (``allow_pickle=True``)."""
    import pickle
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        # (torch's weights_only unpickler raises UnpicklingError for every global it does not allow; a missing or truncated
        # file raises FileNotFoundError / EOFError / RuntimeError and is NOT an invitation to unpickle arbitrary code)
        if not allow_pickle:
            raise RuntimeError('%s is not a plain state_dict checkpoint (%s: %s).  If it is a whole-module pickle written '
                               'by the reference (audiogan.py:936-939) and you trust it, pass allow_pickle=True.'
                               % (path, type(e).__name__, str(e)[:200]))
        return torch.load(path, map_location='cpu', weights_only=False)


def load(prefix, iteration, d=None, g=None, e_g=None, e_d=None, opt_d=None, opt_g=None, strict=True, restore_rng=None,
         allow_pickle=False):
    """``restore_rng``: None = restore the RNG streams when an optimiser is being restored (a resumed training run), leave
    them alone for inference-only loads; True / False force it.  Returns ``extra``; ``load.last_rng`` says what happened
    to the RNG state ('restored', 'cpu only ...', 'not restored')."""
    import warnings
    mods = dict(d=d, g=g, e_g=e_g, e_d=e_d)
    for role, key in _ROLES:
        m = mods[key]
        if m is not None:
            obj = _read(_path(prefix, role, iteration), allow_pickle)
            sd = obj.state_dict() if isinstance(obj, torch.nn.Module) else obj      # reference-style module pickle
            m.load_state_dict(sd, strict=strict)
    extra = None
    load.last_rng = 'not restored'
    p = _path(prefix, 'opt', iteration)
    if os.path.exists(p):
        blob = _read(p, allow_pickle)
        for o, key in ((opt_d, 'opt_d'), (opt_g, 'opt_g')):
            if o is not None and blob.get(key) is not None:
                o.load_state_dict(_to(blob[key], o.params[0].device))
        if restore_rng is None:
            restore_rng = opt_d is not None or opt_g is not None
        if restore_rng:
            if blob.get('torch_rng') is not None:
                torch.set_rng_state(blob['torch_rng'])
                load.last_rng = 'restored'
            cr = blob.get('cuda_rng')
            if cr is not None and torch.cuda.is_available():
                if len(cr) == torch.cuda.device_count():
                    torch.cuda.set_rng_state_all(cr)
                else:
                    # saved on a different number of devices: this process's device continues the stream of the saved
                    # device with the same index (or the first one) - and the caller is told
                    i = torch.cuda.current_device()
                    torch.cuda.set_rng_state(cr[i] if i < len(cr) else cr[0], i)
                    load.last_rng = 'cpu + current device only (checkpoint has %d device streams, this process sees %d)' % (
                        len(cr), torch.cuda.device_count())
                    warnings.warn('audiogan_amd.checkpoint.load: ' + load.last_rng)
        extra = blob.get('extra')
    return extra


load.last_rng = 'not restored'


def _cpu(sd):
    return {k: ([t.detach().cpu() for t in v] if isinstance(v, list) else v) for k, v in sd.items()}


def _to(sd, dev):
    return {k: ([t.to(dev) for t in v] if isinstance(v, list) else v) for k, v in sd.items()}
