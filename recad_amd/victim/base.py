"""Victim base class: the drop-in boundary of the hot path.

Reproduces the contract of the reference's BaseModel/BaseVictim + lazy-init
(recad/model/base.py:21-104, recad/model/victim/base.py:4-7, recad/utils.py:199-269):

* ``Cls.from_config(**kw)`` returns a LAZY object: defaults from ``default.MODEL`` merged
  with the caller's kwargs (unknown keys dropped), no tensors yet;
* every method defined by the victim raises ``NotInstantiatedError`` until ``.I(**kw)``;
* ``.I(**kw)`` returns a NEW, built instance (stored kwargs + kw, typically ``dataset=``),
  carrying ``_init_config`` / ``_model_name``; it is idempotent on a built instance;
* a constructor failure surfaces as ``InstantiateFail``;
* ``reset(**kw)`` -> ``type(self).from_config(**_init_config | kw)``; unknown key -> ValueError.

The mechanism is this build's own (a guard installed by ``__init_subclass__`` and a
``_build`` hook) rather than the reference's class rewriting.
"""
import functools
import logging
from copy import copy

from torch import nn

from ..default import MODEL
from ..utils import InstantiateFail, NotInstantiatedError, get_logger, parse_args

_logger = get_logger(__name__)
_UNGUARDED = {"I", "from_config", "reset", "model_name", "print_help", "info_describe"}


def _guard(fn):
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        if not self.__dict__.get("_is_instantiate", False):
            raise NotInstantiatedError(f"{type(self).__name__}.{fn.__name__} is not enabled since no instantiated")
        return fn(self, *a, **k)

    return wrapper


class BaseVictim(nn.Module):
    #: key into default.MODEL["victim"]; set by subclasses
    victim_name = None
    #: kwargs the caller supplies at .I() time (not in the defaults)
    user_args = "dataset"

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        for name, attr in list(cls.__dict__.items()):
            if name.startswith("__") or name in _UNGUARDED or name == "_build":
                continue
            if isinstance(attr, (classmethod, staticmethod, property)) or not callable(attr):
                continue
            setattr(cls, name, _guard(attr))

    def __init__(self, **config):
        super().__init__()
        self._pending = dict(config)
        self._is_instantiate = False

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_config(cls, **kwargs):
        name = cls.victim_name
        assert name in MODEL["victim"], f"{name} is not on the default victim models"
        defaults = MODEL["victim"][name]
        config = {k: copy(v) for k, v in defaults.items()}
        allowed = set(config) | set(parse_args(cls.user_args)) | set(parse_args(cls.extra_user_args(kwargs)))
        for k, v in kwargs.items():
            if k in allowed:
                config[k] = v
            else:
                _logger.debug(f"Unexpected key [{k}] for {cls}")
        inst = cls(**config)
        inst._init_config = config
        inst._model_name = name
        return inst

    @classmethod
    def extra_user_args(cls, kwargs):
        return ""

    def I(self, **kwargs):
        if self._is_instantiate:
            return self
        config = dict(self._pending)
        config.update(kwargs)
        inst = type(self).__new__(type(self))
        nn.Module.__init__(inst)
        inst._pending = config
        inst._is_instantiate = True
        for k in ("_init_config", "_model_name"):
            if hasattr(self, k):
                setattr(inst, k, getattr(self, k))
        try:
            inst._build(**config)
        except Exception as e:  # same exception type as the reference; the message is kept
            raise InstantiateFail(f"{type(e)}: {e}") from e
        return inst

    def _build(self, **config):
        raise NotImplementedError

    def reset(self, **kwargs):
        if not hasattr(self, "_init_config"):
            raise ValueError("reset method is only for datasets instantiated from_config")
        config = copy(self._init_config)
        for k, v in kwargs.items():
            if k not in config:
                raise ValueError(f"reset arg {k} should be in {list(config)}")
            config[k] = v
        return type(self).from_config(**config)

    # ------------------------------------------------------------------ description
    @property
    def model_name(self):
        return getattr(self, "_model_name", type(self).__name__)

    def info_describe(self):
        return {"input_describe": self.input_describe(), "output_describe": self.output_describe()}

    def print_help(self, **kwargs):
        from pprint import pprint

        info = self.info_describe()
        info["model_name"] = self.model_name
        pprint(info)

    # ------------------------------------------------------------------ shared plumbing
    def _adam_is_fused(self):
        """True when self.optimizer is torch.optim.Adam with the options the fused HIP epilogues implement
        (recad/utils.py:181-183: `optim.Adam(params, lr=...)`, i.e. no amsgrad / weight decay / maximize, ONE
        parameter group).  Anything else runs through `_unfused_epoch` instead of being approximated."""
        import torch
        opt = self.optimizer
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        return (not g.get("amsgrad", False)) and g.get("weight_decay", 0) == 0 and not g.get("maximize", False)

    def _adam_slot(self, p):
        """optimizer.state[p] with the exp_avg / exp_avg_sq / step entries torch.optim.Adam keeps (created on
        first use), so optimizer.state_dict() / load_state_dict() carry the moments of the fused HIP Adam."""
        import torch
        st = self.optimizer.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _unfused_epoch(self, cols, batch, grad_step):
        """Any optimizer the reference's pick_optim can return (recad/utils.py:181-189) other than default Adam:
        per minibatch the HIP path computes the loss and the dense gradients only (grad_step(slices) ->
        (loss_partials row, {param: grad tensor})), the torch optimizer applies them.  Same arithmetic as the
        reference's loss.backward(); optimizer.step() (lightgcn.py:166-168), one host round trip per step."""
        import torch
        n = cols[0].numel()
        losses = []
        for s in range(0, n, batch):
            part, grads = grad_step([c[s:s + batch] for c in cols])
            for p_, g_ in grads.items():
                p_.grad = g_
            self.optimizer.step()
            losses.append(part.reshape(-1).sum().double())
        for p_ in self.parameters():
            p_.grad = None
        return torch.stack(losses).cpu()

    @staticmethod
    def _collect_epoch(dataset, keys):
        """All minibatches of one epoch as three int64 device tensors + the batch size.
        Uses the dataset's ``generate_epoch`` when it has one (this build's dataset), else
        concatenates what ``generate_batch()`` yields (the reference's dataset contract,
        recad/dataset/implicit.py:416-458: equal-size batches, last one short)."""
        import torch

        if hasattr(dataset, "generate_epoch"):
            ep = dataset.generate_epoch()
            return [ep[k] for k in keys], int(ep["batch_size"])
        cols = [[] for _ in keys]
        bs = None
        for dp in dataset.generate_batch():
            if bs is None:
                bs = len(dp[keys[0]])
            for c, k in zip(cols, keys):
                c.append(dp[k].long())
        if bs is None:
            raise ValueError("dataset.generate_batch() yielded no batch")
        return [torch.cat(c).contiguous() for c in cols], bs
