"""Framework registry (reference models/frameworks/__init__.py:1-11)."""


def get_model(args, render_target=None):
    if args.model.framework == "UNISURF":
        raise NotImplementedError          # unreachable in the reference as well (frameworks/__init__.py:2-3)
    if args.model.framework == "NeuS":
        from .neus import get_model as f
    elif args.model.framework == "VolSDF":
        from .volsdf import get_model as f
    else:
        raise NotImplementedError
    return f(args, render_target)
