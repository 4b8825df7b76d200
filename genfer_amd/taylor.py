"""Host-side mirror of the reference's ``TaylorPoly<T>`` surface (src/multivariate_taylor.rs)
over a C-ABI library of opaque polynomial handles.

The class produced by :func:`bind` has the same method names, argument meaning and error
behaviour as the Rust type (``var``, ``subst_var``, ``derivative``, ``shift_down``,
``coefficient`` ... and ``+ - * /``), so parity tests read like the reference's own unit
tests (src/multivariate_taylor.rs:733-1513).  It is generic over *which* library it talks
to: the product binds it to ``libgftaylor.so`` (prefix ``gft_``, HIP kernels on gfx950);
the tests additionally bind it to the CPU oracle (``oracle/liborc.so``, prefix ``orc_``).
Nothing in this module imports or loads the oracle.

Reference panics (``assert!``/``unwrap``; ``panic = "abort"`` in release) surface here as
:class:`TaylorError` carrying ``<prefix>last_error()``.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

USIZE_MAX = 2**64 - 1  # usize::MAX == "untruncated" degree (generating_function.rs:485,569)


class TaylorError(RuntimeError):
    pass


def _sz(seq: Iterable[int]):
    seq = [int(s) for s in seq]
    return (C.c_size_t * max(len(seq), 1))(*seq), len(seq)


_VP = C.c_void_p
_DP = C.POINTER(C.c_double)
_SP = C.POINTER(C.c_size_t)

# name -> (restype, argtypes); the single source of truth for the handle API surface.
# Declared in include/gftaylor.h for the product (prefix gft_/gfti_).
HANDLE_API = {
    "last_error": (C.c_char_p, []),
    "width": (C.c_int, []),
    "from_host": (_VP, [_DP, _SP, _SP, C.c_size_t]),
    "scalar": (_VP, [_DP]),
    "from_u32": (_VP, [C.c_uint32]),
    "zero_with": (_VP, [_SP, C.c_size_t]),
    "var": (_VP, [C.c_size_t, _DP, C.c_size_t]),
    "var_at_zero": (_VP, [C.c_size_t, C.c_size_t]),
    "var_with_degrees_p1": (_VP, [C.c_size_t, _DP, _SP, C.c_size_t]),
    "clone": (_VP, [_VP]),
    "free": (None, [_VP]),
    "num_vars": (C.c_size_t, [_VP]),
    "numel": (C.c_size_t, [_VP]),
    "shape": (None, [_VP, _SP]),
    "degrees_p1": (None, [_VP, _SP]),
    "to_host": (C.c_int, [_VP, _DP]),
    "len_of": (C.c_size_t, [_VP, C.c_size_t]),
    "is_constant": (C.c_int, [_VP]),
    "is_zero": (C.c_int, [_VP]),
    "is_one": (C.c_int, [_VP]),
    "equal": (C.c_int, [_VP, _VP]),
    "format": (C.c_long, [_VP, C.c_int, C.c_char_p, C.c_size_t]),
    "constant_term": (C.c_int, [_VP, _DP]),
    "extract_constant": (C.c_int, [_VP, _DP]),
    "extract_linear": (C.c_int, [_VP, _DP, _DP, _SP]),
    "coefficient": (C.c_int, [_VP, _SP, C.c_size_t, _DP]),
    "add": (_VP, [_VP, _VP]),
    "sub": (_VP, [_VP, _VP]),
    "mul": (_VP, [_VP, _VP]),
    "div": (_VP, [_VP, _VP]),
    "neg": (_VP, [_VP]),
    "add_scaled": (_VP, [_VP, _VP, _DP]),
    "exp": (_VP, [_VP]),
    "log": (_VP, [_VP]),
    "pow": (_VP, [_VP, C.c_uint32]),
    "derivative": (_VP, [_VP, C.c_size_t, C.c_size_t]),
    "taylor_expansion_of_coeff": (_VP, [_VP, C.c_size_t, C.c_size_t]),
    "shift_down": (_VP, [_VP, C.c_size_t, C.c_size_t]),
    "subst_var": (_VP, [_VP, C.c_size_t, _VP]),
    "observe_step": (_VP, [_VP, C.c_size_t, _DP, _DP, C.c_size_t]),
    "derive_scale": (_VP, [_VP, C.c_size_t, _DP, C.c_size_t]),
    "observe_chain": (_VP, [_VP, C.c_size_t, _DP, _DP, C.c_size_t, C.c_size_t]),
    "derivative_truncated": (_VP, [_VP, C.c_size_t, C.c_size_t, C.c_size_t]),
    "coefficients_of_term": (_VP, [_VP, C.c_size_t, C.c_size_t]),
    "taylor_polynomial_terms": (_VP, [_VP, C.c_size_t, _SP, C.c_size_t]),
    "truncate_to_degree_p1": (_VP, [_VP, C.c_size_t]),
    "remove_last_variable": (_VP, [_VP]),
    "extend_to_dim": (_VP, [_VP, C.c_size_t, C.c_size_t]),
    "extend": (_VP, [_VP, _SP, C.c_size_t]),
    "mul_var": (_VP, [_VP, _DP, C.c_size_t, _SP, _SP, C.c_size_t]),
    "mul_linear": (_VP, [_VP, _DP, _DP, C.c_size_t, _SP, _SP, C.c_size_t]),
}


class _Fn:
    """Prefix-bound, signature-checked view of one library."""

    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib, self.prefix = lib, prefix
        for name, (res, args) in HANDLE_API.items():
            f = getattr(lib, prefix + name)  # AttributeError if the symbol is missing: loud
            f.restype, f.argtypes = res, args
            setattr(self, name, f)
        self.W = int(self.width())


def bind(lib: C.CDLL, prefix: str):
    """Create a ``TaylorPoly`` class bound to ``lib``'s ``<prefix>*`` entry points."""
    fn = _Fn(lib, prefix)
    W = fn.W

    def scal(x) -> "C.Array":
        if W == 1:
            return (C.c_double * 1)(float(x))
        if isinstance(x, (tuple, list, np.ndarray)):
            lo, hi = x
        else:
            lo = hi = x
        return (C.c_double * 2)(float(lo), float(hi))

    def unscal(buf):
        return float(buf[0]) if W == 1 else (float(buf[0]), float(buf[1]))

    class TaylorPoly:
        __slots__ = ("_h",)
        _fn = fn
        WIDTH = W

        # ---- lifetime ----------------------------------------------------------------
        def __init__(self, handle):
            if not handle:
                raise TaylorError((fn.last_error() or b"unknown error").decode())
            self._h = handle

        def __del__(self):
            h = getattr(self, "_h", None)
            if h:
                fn.free(h)
                self._h = None

        # ---- constructors (mt:33-46, 208-259, 626-656) -----------------------------
        @classmethod
        def new(cls, coeffs, degrees_p1: Sequence[int]):
            a = np.ascontiguousarray(np.asarray(coeffs, dtype=np.float64))
            shape = a.shape[1:] if W == 2 else a.shape
            if W == 2 and (a.ndim == 0 or a.shape[0] != 2):
                raise ValueError("interval coefficients must be stacked as [2, ...] = (lo, hi)")
            sh, nd = _sz(shape)
            dg, nd2 = _sz(degrees_p1)
            if nd != nd2:
                raise TaylorError("invariant: ndim != degrees_p1.len()")
            return cls(fn.from_host(a.ctypes.data_as(_DP), sh, dg, nd))

        @classmethod
        def from_coeffs(cls, coeffs):
            a = np.asarray(coeffs, dtype=np.float64)
            return cls.new(a, a.shape[1:] if W == 2 else a.shape)

        @classmethod
        def taylor(cls, coeffs, degrees_p1: Optional[Sequence[int]] = None):
            """The reference's ``taylor!`` test macro (mt:658-692)."""
            return cls.from_coeffs(coeffs) if degrees_p1 is None else cls.new(coeffs, degrees_p1)

        @classmethod
        def from_scalar(cls, x):
            return cls(fn.scalar(scal(x)))

        @classmethod
        def from_u32(cls, c: int):
            return cls(fn.from_u32(int(c)))

        @classmethod
        def zero(cls):
            return cls.from_scalar(0.0)

        @classmethod
        def one(cls):
            return cls.from_scalar(1.0)

        @classmethod
        def zero_with(cls, degrees_p1):
            dg, nd = _sz(degrees_p1)
            return cls(fn.zero_with(dg, nd))

        @classmethod
        def var(cls, v: int, x, length: int):
            return cls(fn.var(v, scal(x), length))

        @classmethod
        def var_at_zero(cls, v: int, length: int):
            return cls(fn.var_at_zero(v, length))

        @classmethod
        def var_with_degrees_p1(cls, v: int, x, degrees_p1):
            dg, nd = _sz(degrees_p1)
            return cls(fn.var_with_degrees_p1(v, scal(x), dg, nd))

        def clone(self):
            return type(self)(fn.clone(self._h))

        # ---- queries ------------------------------------------------------------------
        def num_vars(self) -> int:
            return int(fn.num_vars(self._h))

        def degrees_p1(self) -> Tuple[int, ...]:
            n = self.num_vars()
            buf = (C.c_size_t * max(n, 1))()
            fn.degrees_p1(self._h, buf)
            return tuple(int(buf[i]) for i in range(n))

        def shape(self) -> Tuple[int, ...]:
            """Rust ``shape()`` returns the *conceptual* degrees (mt:53-56)."""
            return self.degrees_p1()

        def coeffs_shape(self) -> Tuple[int, ...]:
            n = self.num_vars()
            buf = (C.c_size_t * max(n, 1))()
            fn.shape(self._h, buf)
            return tuple(int(buf[i]) for i in range(n))

        def array(self) -> np.ndarray:
            """Stored (compact) coefficient array; intervals come back stacked [2, ...]."""
            sh = self.coeffs_shape()
            n = int(fn.numel(self._h))
            out = np.empty((W * n,), dtype=np.float64)
            if fn.to_host(self._h, out.ctypes.data_as(_DP)) != 0:
                raise TaylorError((fn.last_error() or b"").decode())
            return out.reshape(((2,) + sh) if W == 2 else sh)

        def is_constant(self) -> bool:
            return bool(fn.is_constant(self._h))

        def is_zero(self) -> bool:
            return bool(fn.is_zero(self._h))

        def is_one(self) -> bool:
            return bool(fn.is_one(self._h))

        def len_of(self, v: int) -> int:
            return int(fn.len_of(self._h, v))

        def constant_term(self):
            buf = (C.c_double * 2)()
            fn.constant_term(self._h, buf)
            return unscal(buf)

        def extract_constant(self):
            buf = (C.c_double * 2)()
            return unscal(buf) if fn.extract_constant(self._h, buf) else None

        def extract_linear(self):
            c, m, v = (C.c_double * 2)(), (C.c_double * 2)(), C.c_size_t()
            if fn.extract_linear(self._h, c, m, C.byref(v)):
                return unscal(c), unscal(m), int(v.value)
            return None

        def coefficient(self, index: Sequence[int]):
            idx, n = _sz(index)
            buf = (C.c_double * 2)()
            if fn.coefficient(self._h, idx, n, buf) != 0:
                raise TaylorError((fn.last_error() or b"").decode())
            return unscal(buf)

        # ---- structure ------------------------------------------------------------------
        def extend_to_dim(self, ndim: int, degree_p1: int):
            return type(self)(fn.extend_to_dim(self._h, ndim, degree_p1))

        def extend(self, new_size):
            ns, n = _sz(new_size)
            return type(self)(fn.extend(self._h, ns, n))

        def remove_last_variable(self):
            return type(self)(fn.remove_last_variable(self._h))

        def truncate_to_degree_p1(self, degree_p1: int):
            return type(self)(fn.truncate_to_degree_p1(self._h, degree_p1))

        def coefficients_of_term(self, v: int, order: int):
            return type(self)(fn.coefficients_of_term(self._h, v, order))

        def taylor_polynomial_terms(self, v: int, orders: Sequence[int]):
            o, n = _sz(orders)
            return type(self)(fn.taylor_polynomial_terms(self._h, v, o, n))

        def derivative(self, v: int, n: int):
            return type(self)(fn.derivative(self._h, v, n))

        def taylor_expansion_of_coeff(self, v: int, n: int):
            return type(self)(fn.taylor_expansion_of_coeff(self._h, v, n))

        def shift_down(self, v: int, n: int):
            return type(self)(fn.shift_down(self._h, v, n))

        def derivative_truncated(self, v: int, n: int, degree_p1: int):
            """Fused derivative(v, n).truncate_to_degree_p1(degree_p1) (the evaluator's Derivative arm)."""
            return type(self)(fn.derivative_truncated(self._h, v, n, degree_p1))

        def observe_step(self, v: int, x, c, degree_p1: int):
            """Fused (derivative(v,1).truncate(d) * var(v,x,d)) * c  (gf.rs:684-689)."""
            return type(self)(fn.observe_step(self._h, v, scal(x), scal(c), degree_p1))

        def observe_chain(self, v: int, x, cs, degree_p1: int):
            """n fused observation steps, innermost first (gf.rs:684-689 as the evaluator unfolds it)."""
            flat = []
            for c in cs:
                flat += list(scal(c))
            buf = (C.c_double * max(len(flat), 1))(*flat)
            return type(self)(fn.observe_chain(self._h, v, scal(x), buf, len(cs), degree_p1))

        def add_scaled(self, other: "TaylorPoly", c):
            """Fused self + other * from(c)  (gf.rs:743-746)."""
            return type(self)(fn.add_scaled(self._h, other._h, scal(c)))

        def derive_scale(self, v: int, c, degree_p1: int):
            """Fused derivative(v,1).truncate(d) * c  (continuous-Poisson observation step, gf.rs:703-706)."""
            return type(self)(fn.derive_scale(self._h, v, scal(c), degree_p1))

        def subst_var(self, v: int, subst: "TaylorPoly"):
            return type(self)(fn.subst_var(self._h, v, subst._h))

        def mul_var(self, m, v: int, shape, degrees_p1):
            sh, n = _sz(shape)
            dg, _ = _sz(degrees_p1)
            return type(self)(fn.mul_var(self._h, scal(m), v, sh, dg, n))

        def mul_linear(self, c, m, v: int, shape, degrees_p1):
            sh, n = _sz(shape)
            dg, _ = _sz(degrees_p1)
            return type(self)(fn.mul_linear(self._h, scal(c), scal(m), v, sh, dg, n))

        # ---- algebra --------------------------------------------------------------------
        def exp(self):
            return type(self)(fn.exp(self._h))

        def log(self):
            return type(self)(fn.log(self._h))

        def pow(self, e: int):
            return type(self)(fn.pow(self._h, int(e)))

        def _coerce(self, o):
            return o if isinstance(o, TaylorPoly) else type(self).from_scalar(o)

        def __add__(self, o):
            return type(self)(fn.add(self._h, self._coerce(o)._h))

        def __sub__(self, o):
            return type(self)(fn.sub(self._h, self._coerce(o)._h))

        def __mul__(self, o):
            return type(self)(fn.mul(self._h, self._coerce(o)._h))

        def __truediv__(self, o):
            return type(self)(fn.div(self._h, self._coerce(o)._h))

        def __neg__(self):
            return type(self)(fn.neg(self._h))

        def __eq__(self, o):
            if not isinstance(o, TaylorPoly):
                return NotImplemented
            return bool(fn.equal(self._h, o._h))

        def __ne__(self, o):
            r = self.__eq__(o)
            return r if r is NotImplemented else not r

        __hash__ = None

        def _format(self, debug: bool) -> str:
            n = fn.format(self._h, int(debug), None, 0)
            if n < 0:
                raise TaylorError((fn.last_error() or b"format failed").decode())
            buf = C.create_string_buffer(n + 1)
            fn.format(self._h, int(debug), buf, n + 1)
            return buf.value.decode()

        def __str__(self):  # impl Display (fmt_polynomial, mt:694-730)
            return self._format(False)

        def __repr__(self):  # impl Debug (mt:632-636)
            return self._format(True)

    TaylorPoly.__qualname__ = f"TaylorPoly[{prefix}]"
    return TaylorPoly
