from .signnet import GIN, MLP, GINConv, GINDeepSigns, MaskedGINDeepSigns, SignNetNodeEncoder

__all__ = ["GIN", "MLP", "GINConv", "GINDeepSigns", "MaskedGINDeepSigns", "SignNetNodeEncoder"]
