"""Message / reduce builtins in the spelling stag uses (`dgl.function`):
copy_edge|copy_e, copy_src|copy_u, u_mul_e, u_add_v; sum, mean, max."""


class Message:
    def __init__(self, kind, *fields):
        self.kind, self.fields = kind, fields

    def __repr__(self):
        return f"{self.kind}{self.fields}"


class Reduce:
    def __init__(self, kind, msg, out):
        self.kind, self.msg, self.out = kind, msg, out

    def __repr__(self):
        return f"{self.kind}({self.msg!r} -> {self.out!r})"


def copy_u(u, out): return Message("copy_u", u, out)
def copy_e(e, out): return Message("copy_e", e, out)
def u_mul_e(u, e, out): return Message("u_mul_e", u, e, out)
def u_add_v(u, v, out): return Message("u_add_v", u, v, out)


copy_src = copy_u      # pre-0.5 DGL names still used by the reference (stag/zoo/gcn.py:59)
copy_edge = copy_e     # stag/layers.py:13


def sum(msg, out): return Reduce("sum", msg, out)       # noqa: A001
def mean(msg, out): return Reduce("mean", msg, out)
def max(msg, out): return Reduce("max", msg, out)       # noqa: A001
