"""Read an rl_games / torch.save checkpoint WITHOUT unpickling it.

The reference ships `results/Bez_Kick/Normal/Bez_Kick_33.pth` (played by its test, test/test_kick_env.py:228-231, through
utils/players.py:68-72).  `torch.load(weights_only=True)` rejects it (the pickle references numpy.core.multiarray.scalar) and
`weights_only=False` would execute whatever an untrusted pickle says.  This reader does neither: `pickletools.genops` only
DISASSEMBLES `archive/data.pkl`; a tiny symbolic stack machine rebuilds the container structure while every GLOBAL stays
an inert name and every REDUCE an inert (name, args) record -- nothing is imported, called or constructed.  Tensor payloads
are the zip's raw little-endian `archive/data/<key>` entries, viewed with numpy.  Unknown opcodes or globals: ValueError.
"""
import pickletools
import zipfile

import numpy as np

_ALLOWED_GLOBALS = {"collections OrderedDict", "torch._utils _rebuild_tensor_v2", "torch FloatStorage", "torch DoubleStorage",
                    "torch HalfStorage", "torch LongStorage", "torch IntStorage", "numpy dtype", "numpy.core.multiarray scalar",
                    "numpy._core.multiarray scalar", "_codecs encode"}
_STORAGE_DTYPE = {"torch FloatStorage": "<f4", "torch DoubleStorage": "<f8", "torch HalfStorage": "<f2",
                  "torch LongStorage": "<i8", "torch IntStorage": "<i4"}


class _Sym:
    def __init__(self, name): self.name = name
    def __repr__(self): return "Sym(%s)" % self.name


class _Call:
    def __init__(self, fn, args): self.fn, self.args, self.state = fn, args, None


class _Pers:
    def __init__(self, pid): self.pid = pid


_MARK = object()


def _symbolic_eval(data):
    stack, memo = [], {}

    def pop_mark():
        i = len(stack) - 1
        while stack[i] is not _MARK:
            i -= 1
        items = stack[i + 1:]
        del stack[i:]
        return items

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n == "PROTO": pass
        elif n == "EMPTY_DICT": stack.append({})
        elif n == "EMPTY_LIST": stack.append([])
        elif n == "EMPTY_TUPLE": stack.append(())
        elif n == "MARK": stack.append(_MARK)
        elif n in ("BINUNICODE", "SHORT_BINUNICODE", "BININT", "BININT1", "BININT2", "BINFLOAT", "LONG1"): stack.append(arg)
        elif n == "NEWTRUE": stack.append(True)
        elif n == "NEWFALSE": stack.append(False)
        elif n == "NONE": stack.append(None)
        elif n == "TUPLE": stack.append(tuple(pop_mark()))
        elif n == "TUPLE1": stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2": stack[-2:] = [(stack[-2], stack[-1])]
        elif n == "TUPLE3": stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n in ("BINPUT", "LONG_BINPUT"): memo[arg] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET"): stack.append(memo[arg])
        elif n == "GLOBAL":
            if arg not in _ALLOWED_GLOBALS:
                raise ValueError("checkpoint references an unexpected global: %r" % (arg,))
            stack.append(_Sym(arg))
        elif n == "REDUCE":
            args = stack.pop(); fn = stack.pop()
            if not isinstance(fn, _Sym):
                raise ValueError("REDUCE on a non-global")
            stack.append({} if fn.name == "collections OrderedDict" else _Call(fn.name, args))  # OrderedDict() -> plain dict
        elif n == "BINPERSID": stack.append(_Pers(stack.pop()))
        elif n == "SETITEM":
            v = stack.pop(); k = stack.pop(); stack[-1][k] = v
        elif n == "SETITEMS":
            items = pop_mark()
            for i in range(0, len(items), 2):
                stack[-1][items[i]] = items[i + 1]
        elif n == "APPEND":
            v = stack.pop(); stack[-1].append(v)
        elif n == "APPENDS":
            items = pop_mark(); stack[-1].extend(items)
        elif n == "BUILD":
            state = stack.pop()
            if isinstance(stack[-1], _Call):
                stack[-1].state = state  # e.g. numpy dtype __setstate__ payload: kept as data, never applied
            elif isinstance(stack[-1], dict) and isinstance(state, dict):
                stack[-1].update(state)
        elif n == "STOP": break
        else:
            raise ValueError("unsupported pickle opcode in checkpoint: %s" % n)
    return stack[-1]


def _materialise(node, zf, prefix):
    if isinstance(node, dict):
        return {k: _materialise(v, zf, prefix) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return type(node)(_materialise(v, zf, prefix) for v in node)
    if isinstance(node, _Call):
        if node.fn == "torch._utils _rebuild_tensor_v2":
            pers, offset, size, stride = node.args[0], node.args[1], node.args[2], node.args[3]
            _tag, styp, key, _dev, numel = pers.pid
            dt = np.dtype(_STORAGE_DTYPE[styp.name])
            buf = zf.read("%s/data/%s" % (prefix, key))
            # every number below comes from the untrusted pickle: validate before building a strided view
            ints = [offset, numel] + list(size) + list(stride)
            if not all(isinstance(v, int) and not isinstance(v, bool) for v in ints):
                raise ValueError("checkpoint tensor record holds non-integer geometry")
            if numel < 0 or numel * dt.itemsize > len(buf) or offset < 0 or len(size) != len(stride):
                raise ValueError("checkpoint tensor record is out of bounds of its storage")
            if any(v < 0 for v in size) or any(v < 0 for v in stride):
                raise ValueError("checkpoint tensor record has a negative size or stride")
            raw = np.frombuffer(buf, dtype=dt, count=numel)
            if len(size) == 0:
                if offset >= numel:
                    raise ValueError("checkpoint scalar offset is out of bounds of its storage")
                return raw[offset].copy()
            if any(v == 0 for v in size):
                return np.zeros(tuple(size), dtype=dt)
            last = offset + sum((sz - 1) * st for sz, st in zip(size, stride))
            if last >= numel:
                raise ValueError("checkpoint tensor view reaches past the end of its storage")
            view = np.lib.stride_tricks.as_strided(raw[offset:], shape=tuple(size), strides=tuple(s * dt.itemsize for s in stride))
            return np.array(view)  # own the memory
        if node.fn in ("numpy.core.multiarray scalar", "numpy._core.multiarray scalar"):  # (dtype record, latin1-encoded raw bytes): decode plain numbers only
            dtc, enc = node.args
            code = dtc.args[0] if isinstance(dtc, _Call) else None
            raw = enc.args[0].encode("latin1") if isinstance(enc, _Call) and isinstance(enc.args[0], str) else None
            if raw is not None and code in ("f8", "f4", "i8", "i4"):
                return np.frombuffer(raw, dtype="<" + code)[0].item()
            return None
        return None
    return node


def read_rlgames_checkpoint(path):
    """-> nested dict with numpy arrays for tensors (model / running_mean_std / reward_mean_std / optimizer ...)."""
    with zipfile.ZipFile(path) as zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")][0]
        prefix = pkl[: -len("/data.pkl")]
        tree = _symbolic_eval(zf.read(pkl))
        return _materialise(tree, zf, prefix)


def load_into_agent_modules(ck, model, running_mean_std=None, value_mean_std=None):
    """Copy the arrays of a checkpoint read by read_rlgames_checkpoint into this build's PPO modules (same key names)."""
    import torch
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ck["model"].items() if isinstance(v, np.ndarray)}
    own = model.state_dict()
    model.load_state_dict({k: sd[k].to(own[k].dtype).reshape(own[k].shape) for k in own})
    for name, mod in (("running_mean_std", running_mean_std), ("reward_mean_std", value_mean_std)):
        if mod is not None and name in ck:
            m = mod.state_dict()
            mod.load_state_dict({k: torch.as_tensor(np.asarray(ck[name][k])).to(m[k].dtype).reshape(m[k].shape) for k in m})
