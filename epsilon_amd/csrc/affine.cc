#include "affine.h"

#include "kernels.h"

namespace eps {

int64_t GetDimension(const pb::Expression& e) {
  EPS_CHECK_MSG(e.size.dim.size() == 2, "expression size must have 2 dims");
  return static_cast<int64_t>(e.size.dim[0]) * e.size.dim[1];
}

void GetVariables(const pb::Expression& e, std::map<std::string, const pb::Expression*>* vars) {
  if (e.expression_type == pb::Expression::VARIABLE) vars->insert(std::make_pair(e.variable_id, &e));
  for (const auto& a : e.arg) GetVariables(a, vars);
}

std::map<std::string, const pb::Expression*> GetVariables(const pb::Problem& p) {
  std::map<std::string, const pb::Expression*> vars;
  GetVariables(p.objective, &vars);
  for (const auto& c : p.constraint) GetVariables(c, &vars);
  return vars;
}

namespace affine {

std::string constraint_key(int i) { return "constraint:" + std::to_string(i); }
std::string arg_key(int i) { return "arg:" + std::to_string(i); }

namespace {

void Impl(const pb::Expression& expr, DataMap* data, const std::string& row_key, LinearMap L,
          BlockMatrix* A, BlockVector* b) {
  switch (expr.expression_type) {
    case pb::Expression::ADD:
    case pb::Expression::RESHAPE:  // affine.cc:30-39, :97 (RESHAPE is a pass-through)
      for (const auto& arg : expr.arg) Impl(arg, data, row_key, L, A, b);
      return;
    case pb::Expression::VARIABLE:  // affine.cc:41-49
      EPS_CHECK_MSG(A != nullptr, "variable in a constant-only affine expression");
      A->InsertOrAdd(row_key, expr.variable_id, L);
      return;
    case pb::Expression::CONSTANT: {  // affine.cc:51-69
      if (b == nullptr) return;
      const pb::Constant& c = data->Resolve(expr.constant);
      DVec b_dense;
      if (c.data_location.empty()) {
        // scalar constant promoted to the width L expects
        b_dense = DVec::Full(L.impl().n(), c.scalar, data->dtype());
      } else {
        b_dense = data->DenseDevice(c);
      }
      b->InsertOrAddApply(row_key, L.impl(), b_dense, 1.0);
      return;
    }
    case pb::Expression::LINEAR_MAP:  // affine.cc:71-84
      EPS_CHECK(expr.arg.size() == 1);
      Impl(expr.arg[0], data, row_key, L * BuildLinearMap(expr.linear_map, data), A, b);
      return;
    default:
      EPS_FATAL("No linear function for expression type " << expr.expression_type);
  }
}

}  // namespace

void BuildAffineOperator(const pb::Expression& expr, DataMap* data, const std::string& row_key,
                         BlockMatrix* A, BlockVector* b) {
  Impl(expr, data, row_key, LinearMap::Identity(GetDimension(expr)), A, b);
}

}  // namespace affine
}  // namespace eps
