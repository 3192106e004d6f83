#include "wire.h"

#include <cstring>

#include "common.h"

namespace eps {
namespace pb {

namespace {

struct Reader {
  const uint8_t* p;
  const uint8_t* end;

  bool done() const { return p >= end; }

  uint64_t varint() {
    uint64_t v = 0;
    int shift = 0;
    while (true) {
      EPS_CHECK_MSG(p < end, "truncated varint");
      uint8_t b = *p++;
      v |= static_cast<uint64_t>(b & 0x7F) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
      EPS_CHECK_MSG(shift < 70, "malformed varint");
    }
    return v;
  }

  double fixed64() {
    EPS_CHECK_MSG(end - p >= 8, "truncated fixed64");
    double d;
    std::memcpy(&d, p, 8);
    p += 8;
    return d;
  }

  Reader sub() {
    uint64_t len = varint();
    EPS_CHECK_MSG(static_cast<uint64_t>(end - p) >= len, "truncated length-delimited field");
    Reader r{p, p + len};
    p += len;
    return r;
  }

  std::string str() {
    Reader r = sub();
    return std::string(reinterpret_cast<const char*>(r.p), r.end - r.p);
  }

  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: EPS_CHECK(end - p >= 8); p += 8; break;
      case 2: sub(); break;
      case 5: EPS_CHECK(end - p >= 4); p += 4; break;
      default: EPS_FATAL("unsupported wire type " << wt);
    }
  }
};

// Returns false at end of message, else field number / wire type.
bool NextField(Reader* r, int* number, int* wt) {
  if (r->done()) return false;
  uint64_t key = r->varint();
  *number = static_cast<int>(key >> 3);
  *wt = static_cast<int>(key & 7);
  return true;
}

int32_t AsInt32(uint64_t v) { return static_cast<int32_t>(static_cast<int64_t>(v)); }

void ParseConstantInto(Reader r, Constant* c) {
  int no, wt;
  while (NextField(&r, &no, &wt)) {
    if (no == 1 && wt == 0) c->constant_type = AsInt32(r.varint());
    else if (no == 2 && wt == 1) c->scalar = r.fixed64();
    else if (no == 3 && wt == 0) c->m = AsInt32(r.varint());
    else if (no == 4 && wt == 0) c->n = AsInt32(r.varint());
    else if (no == 5 && wt == 0) c->nnz = AsInt32(r.varint());
    else if (no == 6 && wt == 2) c->data_location = r.str();
    else if (no == 8 && wt == 2) c->parameter_id = r.str();
    else r.skip(wt);
  }
}

void ParseSizeInto(Reader r, Size* s) {
  int no, wt;
  while (NextField(&r, &no, &wt)) {
    if (no == 1 && wt == 2) {  // packed
      Reader pk = r.sub();
      while (!pk.done()) s->dim.push_back(AsInt32(pk.varint()));
    } else if (no == 1 && wt == 0) {
      s->dim.push_back(AsInt32(r.varint()));
    } else {
      r.skip(wt);
    }
  }
}

void ParseLinearMapInto(Reader r, LinearMap* lm) {
  int no, wt;
  while (NextField(&r, &no, &wt)) {
    if (no == 1 && wt == 0) lm->linear_map_type = AsInt32(r.varint());
    else if (no == 2 && wt == 0) lm->m = AsInt32(r.varint());
    else if (no == 3 && wt == 0) lm->n = AsInt32(r.varint());
    else if (no == 4 && wt == 2) ParseConstantInto(r.sub(), &lm->constant);
    else if (no == 5 && wt == 1) lm->scalar = r.fixed64();
    else if (no == 6 && wt == 2) {
      lm->arg.emplace_back();
      ParseLinearMapInto(r.sub(), &lm->arg.back());
    } else r.skip(wt);
  }
}

void ParseExpressionInto(Reader r, Expression* e);

void ParseProxFunctionInto(Reader r, ProxFunction* f) {
  int no, wt;
  while (NextField(&r, &no, &wt)) {
    if (no == 1 && wt == 0) f->prox_function_type = AsInt32(r.varint());
    else if (no == 2 && wt == 0) f->epigraph = r.varint() != 0;
    else if (no == 3 && wt == 1) f->alpha = r.fixed64();
    else if (no == 4 && wt == 2) {
      f->arg_size.emplace_back();
      ParseSizeInto(r.sub(), &f->arg_size.back());
    } else if (no == 5 && wt == 2) {
      Reader s = r.sub();
      int n2, w2;
      while (NextField(&s, &n2, &w2)) {
        if (n2 == 1 && w2 == 0) f->sum_largest_k = AsInt32(s.varint());
        else s.skip(w2);
      }
    } else if (no == 6 && wt == 2) {
      Reader s = r.sub();
      int n2, w2;
      while (NextField(&s, &n2, &w2)) {
        if (n2 == 1 && w2 == 1) f->sz_alpha = s.fixed64();
        else if (n2 == 2 && w2 == 1) f->sz_beta = s.fixed64();
        else if (n2 == 3 && w2 == 1) f->sz_c = s.fixed64();
        else if (n2 == 4 && w2 == 1) f->sz_m = s.fixed64();
        else if (n2 == 5 && w2 == 2) {
          f->sz_alpha_expr = std::make_shared<Expression>();
          ParseExpressionInto(s.sub(), f->sz_alpha_expr.get());
        } else if (n2 == 6 && w2 == 2) {
          f->sz_beta_expr = std::make_shared<Expression>();
          ParseExpressionInto(s.sub(), f->sz_beta_expr.get());
        } else s.skip(w2);
      }
    } else if (no == 7 && wt == 0) f->has_axis = r.varint() != 0;
    else if (no == 8 && wt == 0) f->axis = AsInt32(r.varint());
    else r.skip(wt);
  }
}

void ParseExpressionInto(Reader r, Expression* e) {
  int no, wt;
  while (NextField(&r, &no, &wt)) {
    if (no == 1 && wt == 0) e->expression_type = AsInt32(r.varint());
    else if (no == 2 && wt == 2) ParseSizeInto(r.sub(), &e->size);
    else if (no == 3 && wt == 2) {
      e->arg.emplace_back();
      ParseExpressionInto(r.sub(), &e->arg.back());
    } else if (no == 8 && wt == 2) ParseConstantInto(r.sub(), &e->constant);
    else if (no == 9 && wt == 2) {
      Reader s = r.sub();
      int n2, w2;
      while (NextField(&s, &n2, &w2)) {
        if (n2 == 1 && w2 == 2) e->variable_id = s.str();
        else s.skip(w2);
      }
    } else if (no == 13 && wt == 2) {
      Reader s = r.sub();
      int n2, w2;
      while (NextField(&s, &n2, &w2)) {
        if (n2 == 1 && w2 == 0) e->cone_type = AsInt32(s.varint());
        else s.skip(w2);
      }
    } else if (no == 18 && wt == 2) ParseLinearMapInto(r.sub(), &e->linear_map);
    else if (no == 19 && wt == 2) ParseProxFunctionInto(r.sub(), &e->prox_function);
    else r.skip(wt);
  }
}

Reader MakeReader(const void* data, size_t len) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  return Reader{p, p + len};
}

// ---- encoder (SolverStatus only) ----------------------------------------------------------

void PutVarint(std::string* out, uint64_t v) {
  while (v >= 0x80) {
    out->push_back(static_cast<char>((v & 0x7F) | 0x80));
    v >>= 7;
  }
  out->push_back(static_cast<char>(v));
}

void PutDouble(std::string* out, int field, double d) {
  if (d == 0) return;  // proto3: default values are not written
  PutVarint(out, (static_cast<uint64_t>(field) << 3) | 1);
  char buf[8];
  std::memcpy(buf, &d, 8);
  out->append(buf, 8);
}

void PutInt(std::string* out, int field, int64_t v) {
  if (v == 0) return;
  PutVarint(out, (static_cast<uint64_t>(field) << 3) | 0);
  PutVarint(out, static_cast<uint64_t>(v));
}

void PutBytes(std::string* out, int field, const std::string& s) {
  if (s.empty()) return;
  PutVarint(out, (static_cast<uint64_t>(field) << 3) | 2);
  PutVarint(out, s.size());
  out->append(s);
}

}  // namespace

Problem ParseProblem(const void* data, size_t len) {
  Problem p;
  Reader r = MakeReader(data, len);
  int no, wt;
  while (NextField(&r, &no, &wt)) {
    if (no == 1 && wt == 2) ParseExpressionInto(r.sub(), &p.objective);
    else if (no == 2 && wt == 2) {
      p.constraint.emplace_back();
      ParseExpressionInto(r.sub(), &p.constraint.back());
    } else r.skip(wt);
  }
  return p;
}

Expression ParseExpression(const void* data, size_t len) {
  Expression e;
  ParseExpressionInto(MakeReader(data, len), &e);
  return e;
}

LinearMap ParseLinearMap(const void* data, size_t len) {
  LinearMap lm;
  ParseLinearMapInto(MakeReader(data, len), &lm);
  return lm;
}

Constant ParseConstant(const void* data, size_t len) {
  Constant c;
  ParseConstantInto(MakeReader(data, len), &c);
  return c;
}

SolverParams ParseSolverParams(const void* data, size_t len) {
  SolverParams sp;
  Reader r = MakeReader(data, len);
  int no, wt;
  while (NextField(&r, &no, &wt)) {
    if (no == 2 && wt == 0) sp.max_iterations = AsInt32(r.varint());
    else if (no == 11 && wt == 1) sp.rho = r.fixed64();
    else if (no == 13 && wt == 1) sp.rel_tol = r.fixed64();
    else if (no == 14 && wt == 1) sp.abs_tol = r.fixed64();
    else if (no == 18 && wt == 0) sp.epoch_iterations = AsInt32(r.varint());
    else if (no == 24 && wt == 0) sp.ignore_stopping_criteria = r.varint() != 0;
    else if (no == 27 && wt == 0) sp.verbose = r.varint() != 0;
    else if (no == 28 && wt == 0) sp.log_iterations = AsInt32(r.varint());
    else if (no == 30 && wt == 0) sp.solver = AsInt32(r.varint());
    else if (no == 31 && wt == 0) sp.warm_start = r.varint() != 0;
    else r.skip(wt);
  }
  return sp;
}

std::string SolverStatus::Serialize() const {
  std::string out;
  PutInt(&out, 1, state);
  PutInt(&out, 3, num_iterations);
  std::string timing;
  PutDouble(&timing, 1, total_time);
  PutDouble(&timing, 2, init_time);
  PutBytes(&out, 4, timing);
  std::string res;
  PutDouble(&res, 1, r_norm);
  PutDouble(&res, 2, s_norm);
  PutDouble(&res, 3, epsilon_primal);
  PutDouble(&res, 4, epsilon_dual);
  PutBytes(&out, 5, res);
  return out;
}

const char* ProxTypeName(int type) {
  switch (type) {
    case ProxFunction::AFFINE: return "AFFINE";
    case ProxFunction::CONSTANT: return "CONSTANT";
    case ProxFunction::ZERO: return "ZERO";
    case ProxFunction::SUM_SQUARE: return "SUM_SQUARE";
    case ProxFunction::NON_NEGATIVE: return "NON_NEGATIVE";
    case ProxFunction::NORM_1: return "NORM_1";
    case ProxFunction::SUM_DEADZONE: return "SUM_DEADZONE";
    case ProxFunction::SUM_EXP: return "SUM_EXP";
    case ProxFunction::SUM_HINGE: return "SUM_HINGE";
    case ProxFunction::SUM_INV_POS: return "SUM_INV_POS";
    case ProxFunction::SUM_KL_DIV: return "SUM_KL_DIV";
    case ProxFunction::SUM_LOGISTIC: return "SUM_LOGISTIC";
    case ProxFunction::SUM_NEG_ENTR: return "SUM_NEG_ENTR";
    case ProxFunction::SUM_NEG_LOG: return "SUM_NEG_LOG";
    case ProxFunction::SUM_QUAD_OVER_LIN: return "SUM_QUAD_OVER_LIN";
    case ProxFunction::SUM_QUANTILE: return "SUM_QUANTILE";
    case ProxFunction::EXP: return "EXP";
    case ProxFunction::LOG_SUM_EXP: return "LOG_SUM_EXP";
    case ProxFunction::MAX: return "MAX";
    case ProxFunction::NORM_2: return "NORM_2";
    case ProxFunction::NORM_INF: return "NORM_INF";
    case ProxFunction::SECOND_ORDER_CONE: return "SECOND_ORDER_CONE";
    case ProxFunction::SUM_LARGEST: return "SUM_LARGEST";
    case ProxFunction::TOTAL_VARIATION_1D: return "TOTAL_VARIATION_1D";
    case ProxFunction::LAMBDA_MAX: return "LAMBDA_MAX";
    case ProxFunction::MATRIX_FRAC: return "MATRIX_FRAC";
    case ProxFunction::NEG_LOG_DET: return "NEG_LOG_DET";
    case ProxFunction::NORM_NUCLEAR: return "NORM_NUCLEAR";
    case ProxFunction::SEMIDEFINITE: return "SEMIDEFINITE";
    case ProxFunction::SIGMA_MAX: return "SIGMA_MAX";
    default: return "UNKNOWN";
  }
}

}  // namespace pb
}  // namespace eps
