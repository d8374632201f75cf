def rand_equation(*a, **k):
    raise NotImplementedError
