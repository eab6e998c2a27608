"""Small helpers shared by the GPU tests."""


def same_table(lines, elines):
    """The printed table, line for line: text equal, or -- where a float computed in float64 on both sides lands on a rounding boundary of its
    printed digits -- every field within one unit of its last printed place."""
    assert len(lines) == len(elines), (len(lines), len(elines))
    for a, b in zip(lines, elines):
        if a == b:
            continue
        fa, fb = a.split(), b.split()
        assert len(fa) == len(fb), (a, b)
        for x, y in zip(fa, fb):
            if x == y:
                continue
            digits = len(x.split(".")[1]) if "." in x else 0
            assert abs(float(x) - float(y)) <= 1.01 * 10.0 ** -digits, (a, b)
