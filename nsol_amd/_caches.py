"""Values derived from a caller's data term and remembered across solver objects.

An outer loop (ADMM, primal-dual with prox_linear_least_squares) builds one
Tikhonov solver per iteration around the same b: b / x_scale, A^T b and |b|^2 are
formed once.  The reference recomputes them on every call
(nsol/proximal_operators.py:117-120, tikhonov_linear_solver.py:146-158), so a
remembered value may only be served while the memory it was derived from is
provably unchanged:

* an entry is tied to the STORAGE objects of its tensors by weak reference -- it is
  dropped the moment that memory is freed, it never keeps a volume alive, and a new
  allocation at the same address is a different storage, so it cannot match;
* within a storage the key holds address, length, dtype and torch's version
  counter.  torch bumps that counter on its own in-place operations; every kernel
  launch of nsol_amd.ops that writes into a tensor bumps it too (ops._wrote);
* memory rewritten by anything torch cannot see -- a caller's own HIP kernel on
  data_ptr(), a DLPack consumer, hipMemcpy through ctypes -- must be followed by
  nsol_amd.invalidate_caches().
"""
import weakref

_registered = []


class DataCache(object):
    """Up to `keep` values keyed on device tensors (and a hashable extra)."""

    def __init__(self, keep):
        self.keep = int(keep)
        self.entries = []                   # (key, storage weakrefs, value)
        _registered.append(self)

    @staticmethod
    def _key(tensors, extra):
        return tuple((t.data_ptr(), t.numel(), str(t.dtype), int(t._version))
                     for t in tensors) + (extra,)

    def lookup(self, tensors, extra=None):
        key = self._key(tensors, extra)
        for k, refs, val in self.entries:
            if k == key and all(r() is t.untyped_storage()
                                for r, t in zip(refs, tensors)):
                return val
        return None

    def store(self, tensors, extra, val):
        refs = tuple(weakref.ref(t.untyped_storage(), self._storage_freed)
                     for t in tensors)
        self.entries.append((self._key(tensors, extra), refs, val))
        del self.entries[:-self.keep]
        return val

    def _storage_freed(self, _):
        self.entries[:] = [e for e in self.entries
                           if all(r() is not None for r in e[1])]

    def clear(self):
        del self.entries[:]


def invalidate_caches():
    """Forget every value derived from a caller's data (b / x_scale, A^T b, |b|^2,
    uploaded host arrays).  Call it after refilling a device tensor behind torch's
    back (see the module docstring); the next solver recomputes from memory."""
    for c in _registered:
        c.clear()
