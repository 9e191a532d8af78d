"""model.from_config("victim", name, **kw): the reference's factory (recad/model/__init__.py:3-21)
for the victims this build implements."""
from . import victim

factories = {"victim": {"lightgcn": victim.LightGCN, "mf": victim.MF, "ncf": victim.NCF}}


def from_config(scope, name, **kwargs):
    return factories[scope][name].from_config(**kwargs)
