"""Registry + config loader of the drop-in boundary.

The reference builds every module from ``projects/configs/*.py`` through mmcv's
``Config``/``Registry`` (tools/train.py:104-136, mmdet3d/models/builder.py:10-85).  mmcv is
not a dependency here; this module provides the subset the unchanged config files need:
python-file configs with ``_base_`` inheritance, attribute-style nested dicts, and a
``build`` that instantiates ``dict(type=...)`` by registered class name.
"""
import copy
import os


class ConfigDict(dict):
    """Nested dict with attribute access (what the reference's constructors expect,
    e.g. ``transformer_decoder.transformerlayers.attn_cfgs.num_heads``,
    mask2former_nusc_occ.py:90)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def wrap(cls, v):
        if isinstance(v, ConfigDict):
            return v
        if isinstance(v, dict):
            return cls(v)
        if isinstance(v, list):
            return [cls.wrap(x) for x in v]
        if isinstance(v, tuple):
            return tuple(cls.wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict.wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def conv(v):
            if isinstance(v, dict):
                return {k: conv(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(conv(x) for x in v)
            return v
        return conv(self)


class Registry:
    def __init__(self, name):
        self.name = name
        self._classes = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._classes and not force and self._classes[key] is not cls:
                raise KeyError(f"{key} already registered in {self.name}")
            self._classes[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._classes.get(key)

    def __contains__(self, key):
        return key in self._classes

    def build(self, cfg, default_args=None):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"{self.name}: cfg must be a dict with a 'type' key, got {cfg!r}")
        args = ConfigDict(copy.deepcopy(dict(cfg)))
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        t = args.pop("type")
        cls = self._classes.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError(f"'{t}' is not registered in the {self.name} registry")
        return cls(**args)


# mmdet / mmdet3d share ONE model tree (mmdet3d/models/builder.py:3-14)
MODELS = Registry("models")
BACKBONES = NECKS = HEADS = DETECTORS = LOSSES = MODELS
ATTENTION = Registry("attention")
POSITIONAL_ENCODING = Registry("positional_encoding")
TRANSFORMER_LAYER = Registry("transformer_layer")
TRANSFORMER_LAYER_SEQUENCE = Registry("transformer_layer_sequence")


def build_model(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d/models/builder.py:75-85 (build_model -> DETECTORS.build)."""
    return MODELS.build(cfg, dict(train_cfg=train_cfg, test_cfg=test_cfg))


# ------------------------------------------------------------------------------ Config
def _merge(base, new):
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and v.pop("_delete_", False):
            out[k] = v
        elif isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = v
    return out


def _exec_file(path):
    scope = {"__file__": path}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), scope)
    return {k: v for k, v in scope.items()
            if not k.startswith("__") and not callable(v) and not isinstance(v, type(os))}


class Config(ConfigDict):
    """``Config.fromfile('projects/configs/occformer_nusc/occformer_nusc_r50_256x704.py')``:
    executes the python config, resolves ``_base_`` files depth-first and merges dicts key
    by key (mmcv.Config semantics incl. ``_delete_=True``)."""

    @staticmethod
    def _load(path):
        cur = _exec_file(path)
        bases = cur.pop("_base_", [])
        if isinstance(bases, str):
            bases = [bases]
        merged = {}
        for b in bases:
            merged = _merge(merged, Config._load(os.path.join(os.path.dirname(path), b)))
        return _merge(merged, cur)

    @classmethod
    def fromfile(cls, path, overrides=None):
        cfg = cls(Config._load(os.path.abspath(path)))
        for dotted, val in (overrides or {}).items():
            node = cfg
            keys = dotted.split(".")
            for k in keys[:-1]:
                node = node[k] if not isinstance(node, list) else node[int(k)]
            node[keys[-1]] = val
        return cfg
