"""Protobuf *wire-format* codec for Epsilon's prox-affine IR (python 3, no protoc).

The reference frontend serialises `Problem` / `Expression` / `SolverParams` with
generated `_pb2` modules (reference: python/epopt/expression.py:33-37,
python/epopt/cvxpy_solver.py:69,91-95) and reads back `SolverStatus`
(cvxpy_solver.py:96).  No `protoc` exists in this image, so the message schemas of
proto/epsilon/{expression,solver,solver_params}.proto are restated here as field
tables and encoded/decoded by hand (varint / fixed64 / length-delimited only).

The bytes produced here are what the C-ABI (`include/epsilon_hip.h`) consumes; the
C++ decoder in `epsilon_amd/csrc/wire.cc` is the other half.  Field numbers are
cited from the .proto files next to each table.
"""

import struct

# wire types
_VARINT, _FIXED64, _LEN, _FIXED32 = 0, 1, 2, 5


def _enc_varint(v):
    if v < 0:
        v += 1 << 64  # int32/int64 negative values are 10-byte varints
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _dec_varint(buf, pos):
    shift = 0
    result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            break
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")
    return result, pos


def _to_signed32(v):
    v &= 0xFFFFFFFFFFFFFFFF
    if v >= 1 << 63:
        v -= 1 << 64
    return int(v)


class Field(object):
    __slots__ = ("name", "number", "kind", "repeated", "msg", "default")

    def __init__(self, name, number, kind, repeated=False, msg=None, default=None):
        self.name = name
        self.number = number
        self.kind = kind  # int | bool | double | string | bytes | message
        self.repeated = repeated
        self.msg = msg
        self.default = default


class Message(object):
    """Tiny schema-driven protobuf message.

    PROTO2 = True keeps explicit presence (only fields that were set are written,
    unset fields read as the declared default) - SolverParams is proto2.
    """

    FIELDS = ()
    PROTO2 = False

    def __init__(self, **kwargs):
        self._set = set()
        for f in self._fields():
            if f.repeated:
                object.__setattr__(self, f.name, [])
            elif f.kind == "message":
                object.__setattr__(self, f.name, None)
            else:
                object.__setattr__(self, f.name, self._scalar_default(f))
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def _fields(cls):
        return cls.FIELDS

    @staticmethod
    def _scalar_default(f):
        if f.default is not None:
            return f.default
        return {"int": 0, "bool": False, "double": 0.0, "string": "", "bytes": b""}[f.kind]

    def __setattr__(self, name, value):
        if name != "_set":
            names = [f.name for f in self._fields()]
            if name not in names:
                raise AttributeError("%s has no field %r" % (type(self).__name__, name))
            for f in self._fields():
                if f.name == name and f.repeated:
                    value = list(value)
            self._set.add(name)
        object.__setattr__(self, name, value)

    def has(self, name):
        return name in self._set

    # ---- field sub-message helpers -------------------------------------------------
    def _msgclass(self, f):
        m = f.msg
        if isinstance(m, str):
            m = _REGISTRY[m]
        return m

    # ---- encode ---------------------------------------------------------------------
    def SerializeToString(self):
        out = bytearray()
        for f in self._fields():
            v = getattr(self, f.name)
            if f.repeated:
                if not v:
                    continue
                if f.kind == "int":  # packed (proto3 default for scalar numerics)
                    payload = b"".join(_enc_varint(int(x)) for x in v)
                    out += _enc_varint((f.number << 3) | _LEN) + _enc_varint(len(payload)) + payload
                elif f.kind == "double":
                    payload = b"".join(struct.pack("<d", float(x)) for x in v)
                    out += _enc_varint((f.number << 3) | _LEN) + _enc_varint(len(payload)) + payload
                else:
                    for x in v:
                        out += self._enc_one(f, x)
                continue
            if f.kind == "message":
                if v is not None:
                    out += self._enc_one(f, v)
                continue
            if self.PROTO2:
                if f.name in self._set:
                    out += self._enc_one(f, v)
            else:
                if v != self._zero(f):
                    out += self._enc_one(f, v)
        return bytes(out)

    @staticmethod
    def _zero(f):
        return {"int": 0, "bool": False, "double": 0.0, "string": "", "bytes": b""}[f.kind]

    def _enc_one(self, f, v):
        if f.kind == "int":
            return _enc_varint((f.number << 3) | _VARINT) + _enc_varint(int(v))
        if f.kind == "bool":
            return _enc_varint((f.number << 3) | _VARINT) + _enc_varint(1 if v else 0)
        if f.kind == "double":
            return _enc_varint((f.number << 3) | _FIXED64) + struct.pack("<d", float(v))
        if f.kind == "string":
            b = v.encode("utf-8")
            return _enc_varint((f.number << 3) | _LEN) + _enc_varint(len(b)) + b
        if f.kind == "bytes":
            return _enc_varint((f.number << 3) | _LEN) + _enc_varint(len(v)) + bytes(v)
        if f.kind == "message":
            b = v.SerializeToString()
            return _enc_varint((f.number << 3) | _LEN) + _enc_varint(len(b)) + b
        raise ValueError(f.kind)

    # ---- decode ---------------------------------------------------------------------
    @classmethod
    def FromString(cls, data):
        self = cls()
        self._set = set()
        byno = {f.number: f for f in cls._fields()}
        buf = memoryview(bytes(data))
        pos, end = 0, len(buf)
        while pos < end:
            key, pos = _dec_varint(buf, pos)
            number, wt = key >> 3, key & 7
            f = byno.get(number)
            if wt == _VARINT:
                raw, pos = _dec_varint(buf, pos)
                val = raw
            elif wt == _FIXED64:
                val = bytes(buf[pos:pos + 8])
                pos += 8
            elif wt == _LEN:
                ln, pos = _dec_varint(buf, pos)
                val = bytes(buf[pos:pos + ln])
                if len(val) != ln:
                    raise ValueError("truncated message")
                pos += ln
            elif wt == _FIXED32:
                val = bytes(buf[pos:pos + 4])
                pos += 4
            else:
                raise ValueError("unsupported wire type %d" % wt)
            if f is None:
                continue  # unknown field: skip
            self._store(f, wt, val)
        return self

    def _store(self, f, wt, val):
        if f.kind == "int":
            if wt == _LEN:  # packed
                p = 0
                mv = memoryview(val)
                while p < len(mv):
                    x, p = _dec_varint(mv, p)
                    getattr(self, f.name).append(_to_signed32(x))
                self._set.add(f.name)
                return
            v = _to_signed32(val)
        elif f.kind == "bool":
            v = bool(val)
        elif f.kind == "double":
            if wt == _LEN and f.repeated:
                for i in range(0, len(val), 8):
                    getattr(self, f.name).append(struct.unpack("<d", val[i:i + 8])[0])
                self._set.add(f.name)
                return
            v = struct.unpack("<d", val)[0]
        elif f.kind == "string":
            v = val.decode("utf-8")
        elif f.kind == "bytes":
            v = val
        elif f.kind == "message":
            v = self._msgclass(f).FromString(val)
        else:
            raise ValueError(f.kind)
        if f.repeated:
            getattr(self, f.name).append(v)
            self._set.add(f.name)
        else:
            object.__setattr__(self, f.name, v)
            self._set.add(f.name)

    def __repr__(self):
        parts = []
        for f in self._fields():
            v = getattr(self, f.name)
            if f.repeated and not v:
                continue
            if v is None:
                continue
            if not f.repeated and f.kind != "message" and v == self._zero(f) and not self.PROTO2:
                continue
            parts.append("%s=%r" % (f.name, v))
        return "%s(%s)" % (type(self).__name__, ", ".join(parts))

    def __eq__(self, other):
        return type(self) is type(other) and self.SerializeToString() == other.SerializeToString()


_REGISTRY = {}


def _register(cls):
    _REGISTRY[cls.__name__] = cls
    return cls


# ---- proto/epsilon/expression.proto ----------------------------------------------------


@_register
class Constant(Message):  # expression.proto:4-25
    UNKNOWN, DENSE_MATRIX, SPARSE_MATRIX, SCALAR = 0, 1, 2, 3
    FIELDS = (
        Field("constant_type", 1, "int"),
        Field("scalar", 2, "double"),
        Field("m", 3, "int"),
        Field("n", 4, "int"),
        Field("nnz", 5, "int"),
        Field("data_location", 6, "string"),
        Field("data_value", 7, "bytes"),
        Field("parameter_id", 8, "string"),
    )


@_register
class Variable(Message):  # expression.proto:27-29
    FIELDS = (Field("variable_id", 1, "string"),)


@_register
class Size(Message):  # expression.proto:31-33
    FIELDS = (Field("dim", 1, "int", repeated=True),)


@_register
class Cone(Message):  # expression.proto:81-92
    UNKNOWN, ZERO, NON_NEGATIVE, SECOND_ORDER, EXPONENTIAL, SEMIDEFINITE = range(6)
    FIELDS = (Field("cone_type", 1, "int"),)


@_register
class LinearMap(Message):  # expression.proto:94-120
    UNKNOWN, DENSE_MATRIX, SPARSE_MATRIX, DIAGONAL_MATRIX, SCALAR, KRONECKER_PRODUCT, TRANSPOSE = range(7)
    FIELDS = (
        Field("linear_map_type", 1, "int"),
        Field("m", 2, "int"),
        Field("n", 3, "int"),
        Field("constant", 4, "message", msg="Constant"),
        Field("scalar", 5, "double"),
        Field("arg", 6, "message", repeated=True, msg="LinearMap"),
    )


@_register
class SumLargestParams(Message):  # expression.proto:174-177
    FIELDS = (Field("k", 1, "int"),)


@_register
class ProxScaledZoneParams(Message):  # expression.proto:180-192
    FIELDS = (
        Field("alpha", 1, "double"),
        Field("beta", 2, "double"),
        Field("c", 3, "double"),
        Field("m", 4, "double"),
        Field("alpha_expr", 5, "message", msg="Expression"),
        Field("beta_expr", 6, "message", msg="Expression"),
    )


@_register
class ProxFunction(Message):  # expression.proto:122-197
    UNKNOWN = 0
    AFFINE = 1
    CONSTANT = 2
    ZERO = 10
    SUM_SQUARE = 11
    NON_NEGATIVE = 20
    NORM_1 = 21
    SUM_DEADZONE = 22
    SUM_EXP = 23
    SUM_HINGE = 24
    SUM_INV_POS = 25
    SUM_KL_DIV = 26
    SUM_LOGISTIC = 27
    SUM_NEG_ENTR = 28
    SUM_NEG_LOG = 29
    SUM_QUAD_OVER_LIN = 30
    SUM_QUANTILE = 31
    EXP = 32
    LOG_SUM_EXP = 100
    MAX = 101
    NORM_2 = 102
    NORM_INF = 103
    SECOND_ORDER_CONE = 104
    SUM_LARGEST = 105
    TOTAL_VARIATION_1D = 106
    LAMBDA_MAX = 200
    MATRIX_FRAC = 201
    NEG_LOG_DET = 202
    NORM_NUCLEAR = 203
    SEMIDEFINITE = 204
    SIGMA_MAX = 205
    FIELDS = (
        Field("prox_function_type", 1, "int"),
        Field("epigraph", 2, "bool"),
        Field("alpha", 3, "double"),
        Field("arg_size", 4, "message", repeated=True, msg="Size"),
        Field("sum_largest_params", 5, "message", msg="SumLargestParams"),
        Field("scaled_zone_params", 6, "message", msg="ProxScaledZoneParams"),
        Field("has_axis", 7, "bool"),
        Field("axis", 8, "int"),
    )

    @classmethod
    def type_name(cls, value):
        for k, v in vars(cls).items():
            if k.isupper() and isinstance(v, int) and v == value:
                return k
        return str(value)


@_register
class Expression(Message):  # expression.proto:205-334 (solver-visible fields only)
    UNKNOWN = 0
    INDICATOR = 1
    CONSTANT = 2
    VARIABLE = 3
    ADD = 10
    RESHAPE = 25
    LINEAR_MAP = 300
    PROX_FUNCTION = 301
    FIELDS = (
        Field("expression_type", 1, "int"),
        Field("size", 2, "message", msg="Size"),
        Field("arg", 3, "message", repeated=True, msg="Expression"),
        Field("constant", 8, "message", msg="Constant"),
        Field("variable", 9, "message", msg="Variable"),
        Field("cone", 13, "message", msg="Cone"),
        Field("linear_map", 18, "message", msg="LinearMap"),
        Field("prox_function", 19, "message", msg="ProxFunction"),
    )


@_register
class Problem(Message):  # expression.proto:339-346
    FIELDS = (
        Field("objective", 1, "message", msg="Expression"),
        Field("constraint", 2, "message", repeated=True, msg="Expression"),
    )


# ---- proto/epsilon/solver_params.proto (proto2) ----------------------------------------


@_register
class SolverParams(Message):  # solver_params.proto:4-71 (fields the solver reads)
    PROTO2 = True
    PROX_ADMM, PROX_ADMM_TWO_BLOCK = 0, 1
    FIELDS = (
        Field("max_iterations", 2, "int", default=10000),
        Field("rho", 11, "double", default=1.0),
        Field("rel_tol", 13, "double", default=1e-2),
        Field("abs_tol", 14, "double", default=1e-4),
        Field("epoch_iterations", 18, "int", default=10),
        Field("ignore_stopping_criteria", 24, "bool", default=False),
        Field("verbose", 27, "bool", default=False),
        Field("log_iterations", 28, "int", default=100),
        Field("use_epigraph", 29, "bool", default=True),
        Field("solver", 30, "int", default=0),
        Field("warm_start", 31, "bool", default=False),
        Field("warm_start_key", 32, "string", default=""),
    )


# ---- proto/epsilon/solver.proto ---------------------------------------------------------


@_register
class Timing(Message):  # solver.proto:24-32
    FIELDS = (
        Field("total_time", 1, "double"),
        Field("init_time", 2, "double"),
    )


@_register
class Residuals(Message):  # solver.proto:38-46
    FIELDS = (
        Field("r_norm", 1, "double"),
        Field("s_norm", 2, "double"),
        Field("epsilon_primal", 3, "double"),
        Field("epsilon_dual", 4, "double"),
        Field("x_norm", 5, "double"),
        Field("y_norm", 6, "double"),
    )


@_register
class SolverStatus(Message):  # solver.proto:4-60
    NOT_STARTED, INITIALIZING, RUNNING, OPTIMAL, MAX_ITERATIONS_REACHED, ERROR = range(6)
    FIELDS = (
        Field("state", 1, "int"),
        Field("objective_value", 2, "double"),
        Field("num_iterations", 3, "int"),
        Field("timing", 4, "message", msg="Timing"),
        Field("residuals", 5, "message", msg="Residuals"),
    )
