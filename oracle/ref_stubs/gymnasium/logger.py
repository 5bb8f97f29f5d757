def warn(*_a, **_k):
    pass
