"""Host logic without a GPU: the py3 wire codec against the real protobuf runtime, and the
C-ABI library (loads, exports every symbol include/epsilon_hip.h declares, fails loudly
without a device)."""

import ctypes
import os
import re

import numpy as np
import pytest

from epsilon_amd import _solve, ir, problems, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roundtrip_problem():
    prob, _ = problems.lasso(5, 20, seed=0)
    b = prob.SerializeToString()
    p = wire.Problem.FromString(b)
    assert p.SerializeToString() == b
    assert p.objective.expression_type == wire.Expression.ADD
    assert [t.prox_function.prox_function_type for t in p.objective.arg] == [wire.ProxFunction.SUM_SQUARE, wire.ProxFunction.NORM_1]
    assert p.constraint[0].cone.cone_type == wire.Cone.ZERO
    assert list(p.objective.arg[0].arg[0].arg[0].size.dim) == [5, 1]


def test_solver_params_proto2_defaults():
    sp = wire.SolverParams.FromString(b"")
    assert (sp.max_iterations, sp.rho, sp.rel_tol, sp.abs_tol, sp.epoch_iterations) == (10000, 1.0, 1e-2, 1e-4, 10)
    assert wire.SolverParams().SerializeToString() == b""  # unset proto2 fields are not written
    sp = wire.SolverParams.FromString(wire.SolverParams(solver=1, max_iterations=7).SerializeToString())
    assert sp.solver == 1 and sp.max_iterations == 7 and sp.rel_tol == 1e-2


def test_against_google_protobuf_runtime():
    """Encode Constant / Size / LinearMap with the official python protobuf runtime (dynamic
    descriptors restating proto/epsilon/expression.proto) and compare bytes both ways."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="t_expr.proto", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto
    c = fd.message_type.add(name="Constant")
    for name, num, typ in [("constant_type", 1, T.TYPE_INT32), ("scalar", 2, T.TYPE_DOUBLE), ("m", 3, T.TYPE_INT32),
                           ("n", 4, T.TYPE_INT32), ("nnz", 5, T.TYPE_INT32), ("data_location", 6, T.TYPE_STRING),
                           ("parameter_id", 8, T.TYPE_STRING)]:
        c.field.add(name=name, number=num, type=typ, label=T.LABEL_OPTIONAL)
    s = fd.message_type.add(name="Size")
    s.field.add(name="dim", number=1, type=T.TYPE_INT32, label=T.LABEL_REPEATED)
    lm = fd.message_type.add(name="LinearMap")
    for name, num, typ in [("linear_map_type", 1, T.TYPE_INT32), ("m", 2, T.TYPE_INT32), ("n", 3, T.TYPE_INT32),
                           ("scalar", 5, T.TYPE_DOUBLE)]:
        lm.field.add(name=name, number=num, type=typ, label=T.LABEL_OPTIONAL)
    lm.field.add(name="constant", number=4, type=T.TYPE_MESSAGE, type_name=".Constant", label=T.LABEL_OPTIONAL)
    lm.field.add(name="arg", number=6, type=T.TYPE_MESSAGE, type_name=".LinearMap", label=T.LABEL_REPEATED)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        pytest.skip("protobuf runtime without GetMessageClass")
    PConstant, PSize, PLM = (get(pool.FindMessageTypeByName(n)) for n in ("Constant", "Size", "LinearMap"))
    g = PConstant(constant_type=1, m=7, n=-3, scalar=2.5, data_location="/mem/data/abc")
    mine = wire.Constant(constant_type=1, m=7, n=-3, scalar=2.5, data_location="/mem/data/abc")
    assert mine.SerializeToString() == g.SerializeToString()
    assert wire.Constant.FromString(g.SerializeToString()) == mine
    assert wire.Size(dim=[10000, 1]).SerializeToString() == PSize(dim=[10000, 1]).SerializeToString()
    gk = PLM(linear_map_type=5, m=6, n=6, arg=[PLM(linear_map_type=4, m=2, n=2, scalar=-1.0),
                                                 PLM(linear_map_type=1, m=3, n=3, constant=g)])
    mk = wire.LinearMap(linear_map_type=5, m=6, n=6,
                        arg=[wire.LinearMap(linear_map_type=4, m=2, n=2, scalar=-1.0),
                             wire.LinearMap(linear_map_type=1, m=3, n=3, constant=mine)])
    assert mk.SerializeToString() == gk.SerializeToString()


def test_constant_packing_matches_reference_layout():
    """python/epopt/constant.py:10-34: dense = float64 column-major; sparse = CSC
    indptr|indices|data as int32,int32,float64."""
    import scipy.sparse as sp
    A = np.arange(6, dtype=np.float64).reshape(2, 3)
    c, b = ir.value_data(A)
    assert (c.m, c.n) == (2, 3) and np.array_equal(np.frombuffer(b), [0, 3, 1, 4, 2, 5])
    S = sp.csc_matrix(np.array([[0, 2.0], [3.0, 0]]))
    c, b = ir.value_data(S)
    assert c.nnz == 2 and len(b) == 2 * 8 + (2 + 2 + 1) * 4


def test_library_exports_every_declared_symbol():
    from epsilon_amd import build
    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, "include", "epsilon_hip.h")).read()
    names = set(re.findall(r"\b(eps_[a-z0-9_]+)\s*\(", header))
    names -= {"eps_allreduce_fn"}
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), "libepsilon_hip.so does not export %s" % n


def test_no_cpu_fallback_without_device():
    """The product path must fail loudly when there is no HIP device (no CPU fallback)."""
    if _solve.device_count() > 0:
        pytest.skip("a HIP device is present")
    prob, _ = problems.lasso(5, 20, seed=0)
    with pytest.raises(_solve.error) as e:
        _solve.solve(prob.SerializeToString(), [], b"", prob.expression_data())
    assert "no HIP device" in str(e.value)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "epsilon_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("the oracle", ""), f
