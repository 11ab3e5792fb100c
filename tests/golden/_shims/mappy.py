"""Import shim (fixture generation only): minimal stand-in for the parts of `mappy`
that are touched when the reference's RUNS path is imported. Only FASTA reading is real."""
import gzip


def fastx_read(path, read_comment=False):
    opener = gzip.open if str(path).endswith(".gz") else open
    name, chunks = None, []
    with opener(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks), None
                name, chunks = line[1:].split()[0], []
            elif line:
                chunks.append(line)
    if name is not None:
        yield name, "".join(chunks), None


class Aligner:
    def __init__(self, *a, **k):
        pass

    def map(self, *a, **k):
        return iter(())


class ThreadBuffer:
    pass
