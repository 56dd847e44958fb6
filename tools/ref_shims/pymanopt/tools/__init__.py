import collections


def make_enum(name, fields):
    """pymanopt 0.2.5: a namedtuple instance whose fields hold 0..len-1 (indexable, attributes)."""
    return collections.namedtuple(name, fields)(*range(len(fields)))
