def make_enum(name, fields):
    class _Enum(list):
        pass

    e = _Enum(fields)
    for i, f in enumerate(fields):
        setattr(e, f, i)
    e.__name__ = name
    return e
