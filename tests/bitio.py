"""MSB-first bit writer for hand-made BVGraph streams (test helper; code definitions: SURVEY.md App. B)."""


class BitWriter:
    def __init__(self):
        self.bits = []  # list of '0'/'1' chunks

    def __len__(self):
        return sum(len(b) for b in self.bits)

    def raw(self, s):
        self.bits.append(s)

    def unary(self, x):
        self.bits.append("0" * x + "1")

    def gamma(self, x):
        v = x + 1
        m = v.bit_length() - 1
        self.bits.append("0" * m + "1" + (format(v - (1 << m), "0%db" % m) if m else ""))

    def zeta(self, x, k=3):
        v = x + 1
        msb = v.bit_length() - 1
        h = msb // k
        self.bits.append("0" * h + "1")
        left = 1 << (h * k)
        if v - left < left:  # short codeword: h*k + k - 1 bits
            self.bits.append(format(v - left, "0%db" % (h * k + k - 1)) if h * k + k - 1 else "")
        else:
            self.bits.append(format(v, "0%db" % (h * k + k)))

    def tobytes(self):
        s = "".join(self.bits)
        s += "0" * (-len(s) % 8)
        return int(s, 2).to_bytes(len(s) // 8, "big") if s else b""


def int2nat(x):
    return 2 * x if x >= 0 else -2 * x - 1


def write_graph(basename, records, window=7, min_interval=0, zeta_k=3, max_ref_count=3, arcs=0):
    """records: list of callables f(BitWriter) each writing one node's record; writes .graph/.offsets/.properties."""
    w = BitWriter()
    offs = BitWriter()
    prev = 0
    offs.gamma(0)
    for rec in records:
        rec(w)
        offs.gamma(len(w) - prev)
        prev = len(w)
    open(basename + ".graph", "wb").write(w.tobytes())
    open(basename + ".offsets", "wb").write(offs.tobytes())
    open(basename + ".properties", "w").write(
        "graphclass=it.unimi.dsi.webgraph.BVGraph\nversion=0\nnodes=%d\narcs=%d\nwindowsize=%d\nmaxrefcount=%d\n"
        "minintervallength=%d\nzetak=%d\ncompressionflags=\n" % (len(records), arcs, window, max_ref_count, min_interval, zeta_k))
