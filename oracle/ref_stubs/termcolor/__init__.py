def colored(s, *_a, **_k):
    return s
