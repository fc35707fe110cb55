"""Kaldi ark/scp matrix I/O — the wire format the reference's decode loop hands to the vocoder
(`kaldiio.WriteHelper("ark,scp:{o}.ark,{o}.scp")`, tts.py:652,674; consumed by `parallel-wavegan-decode --feats-scp`,
inference_student.sh:20-23).  kaldiio is not a dependency here: binary float32 matrices are written directly.

Record layout (Kaldi binary FloatMatrix):  <key> ' ' '\\0' 'B' 'F' 'M' ' ' '\\x04' <int32 rows> '\\x04' <int32 cols> <rows*cols float32>
scp line:                                  <key> ' ' <ark path> ':' <byte offset of the '\\0B' marker>
"""
import struct

import numpy as np


class ArkScpWriter(object):
    def __init__(self, prefix):
        self.ark_path, self.scp_path = prefix + ".ark", prefix + ".scp"
        self._ark = open(self.ark_path, "wb")
        self._scp = open(self.scp_path, "w")

    def __setitem__(self, key, mat):
        mat = np.ascontiguousarray(mat, dtype=np.float32)
        assert mat.ndim == 2 and " " not in key
        self._ark.write(key.encode("utf-8") + b" ")
        off = self._ark.tell()
        self._ark.write(b"\0BFM " + b"\x04" + struct.pack("<i", mat.shape[0]) + b"\x04" + struct.pack("<i", mat.shape[1]))
        self._ark.write(mat.tobytes())
        self._scp.write("%s %s:%d\n" % (key, self.ark_path, off))

    def write_batch(self, keys, mats, counts):
        """The mels of a decoded batch: `mats` float32 [sum(counts), cols] holds them back to back (the pinned landing buffer), counts[i] rows each.
        One native call (fcl_kaldi_ark_append: writev straight from `mats`) and one scp write per batch instead of three writes and a copy per
        utterance."""
        import ctypes as C

        from . import _lib

        mats = np.ascontiguousarray(mats, dtype=np.float32)
        n = len(keys)
        assert mats.ndim == 2 and len(counts) == n and int(sum(counts)) == mats.shape[0] and all(" " not in k for k in keys)
        self._ark.flush()
        pos = self._ark.tell()
        kb = [k.encode("utf-8") for k in keys]
        karr = (C.c_char_p * n)(*kb)
        rows = np.ascontiguousarray(counts, dtype=np.int32)
        offs = np.zeros(n, dtype=np.int64)
        end = _lib.load().fcl_kaldi_ark_append(self._ark.fileno(), pos, n, karr, mats.ctypes.data, rows.ctypes.data, int(mats.shape[1]), offs.ctypes.data)
        if end < 0:
            raise _lib.FclError(_lib.load().fcl_last_error().decode())
        self._ark.seek(end)
        self._scp.write("".join("%s %s:%d\n" % (k, self.ark_path, o) for k, o in zip(keys, offs.tolist())))

    def close(self):
        self._ark.close()
        self._scp.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_mat(ark_path, offset):
    with open(ark_path, "rb") as f:
        f.seek(offset)
        assert f.read(2) == b"\0B" and f.read(3) == b"FM "
        assert f.read(1) == b"\x04"
        rows = struct.unpack("<i", f.read(4))[0]
        assert f.read(1) == b"\x04"
        cols = struct.unpack("<i", f.read(4))[0]
        return np.frombuffer(f.read(4 * rows * cols), dtype=np.float32).reshape(rows, cols).copy()


def read_vec(ark_path, offset):
    """A Kaldi binary FloatVector ('\\0B' 'FV ' '\\x04' <int32 dim> <float32 data>) -- the x-vector of an utterance as the reference's data.json points
    at it (`input[1].feat = "<ark>:<offset>"`, tts.py:330, 285-287); a 1 x dim or dim x 1 FloatMatrix is accepted too."""
    with open(ark_path, "rb") as f:
        f.seek(offset)
        assert f.read(2) == b"\0B"
        kind = f.read(3)
        if kind == b"FM ":
            return read_mat(ark_path, offset).reshape(-1)
        assert kind == b"FV " and f.read(1) == b"\x04"
        dim = struct.unpack("<i", f.read(4))[0]
        return np.frombuffer(f.read(4 * dim), dtype=np.float32).copy()


def read_scp(scp_path):
    """{key: matrix} for every line of an scp file written by ArkScpWriter (or by Kaldi/kaldiio for float matrices)."""
    out = {}
    with open(scp_path) as f:
        for line in f:
            key, loc = line.strip().split(None, 1)
            path, off = loc.rsplit(":", 1)
            out[key] = read_mat(path, int(off))
    return out
