// Hand-written protobuf wire decoder for the three Epsilon .proto files.
//
// The reference links libprotobuf and generated *.pb.h (reference Makefile:46,
// proto/epsilon/{expression,solver,solver_params}.proto); neither exists in this image, so
// the messages the solver reads are restated as plain structs and decoded from the wire
// (varint / fixed64 / length-delimited).  Field numbers are cited per struct.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace eps {
namespace pb {

struct Constant {  // expression.proto:4-25
  enum Type { UNKNOWN = 0, DENSE_MATRIX = 1, SPARSE_MATRIX = 2, SCALAR = 3 };
  int constant_type = 0;       // 1
  double scalar = 0;           // 2
  int32_t m = 0, n = 0, nnz = 0;  // 3,4,5
  std::string data_location;   // 6
  std::string parameter_id;    // 8
};

struct Size {  // expression.proto:31-33
  std::vector<int32_t> dim;  // 1 (packed or not)
};

struct LinearMap {  // expression.proto:94-120
  enum Type {
    UNKNOWN = 0, DENSE_MATRIX = 1, SPARSE_MATRIX = 2, DIAGONAL_MATRIX = 3, SCALAR = 4,
    KRONECKER_PRODUCT = 5, TRANSPOSE = 6
  };
  int linear_map_type = 0;  // 1
  int32_t m = 0, n = 0;     // 2,3
  Constant constant;        // 4
  double scalar = 0;        // 5
  std::vector<LinearMap> arg;  // 6
};

struct Expression;

struct ProxFunction {  // expression.proto:122-197
  enum Type {
    UNKNOWN = 0, AFFINE = 1, CONSTANT = 2, ZERO = 10, SUM_SQUARE = 11, NON_NEGATIVE = 20,
    NORM_1 = 21, SUM_DEADZONE = 22, SUM_EXP = 23, SUM_HINGE = 24, SUM_INV_POS = 25,
    SUM_KL_DIV = 26, SUM_LOGISTIC = 27, SUM_NEG_ENTR = 28, SUM_NEG_LOG = 29,
    SUM_QUAD_OVER_LIN = 30, SUM_QUANTILE = 31, EXP = 32, LOG_SUM_EXP = 100, MAX = 101,
    NORM_2 = 102, NORM_INF = 103, SECOND_ORDER_CONE = 104, SUM_LARGEST = 105,
    TOTAL_VARIATION_1D = 106, LAMBDA_MAX = 200, MATRIX_FRAC = 201, NEG_LOG_DET = 202,
    NORM_NUCLEAR = 203, SEMIDEFINITE = 204, SIGMA_MAX = 205
  };
  int prox_function_type = 0;  // 1
  bool epigraph = false;       // 2
  double alpha = 0;            // 3
  std::vector<Size> arg_size;  // 4
  int32_t sum_largest_k = 0;   // 5{1}
  // scaled_zone_params (6): 1 alpha, 2 beta, 3 c, 4 m, 5 alpha_expr, 6 beta_expr
  double sz_alpha = 0, sz_beta = 0, sz_c = 0, sz_m = 0;
  std::shared_ptr<Expression> sz_alpha_expr, sz_beta_expr;
  bool has_axis = false;  // 7
  int32_t axis = 0;       // 8
};

const char* ProxTypeName(int type);

struct Expression {  // expression.proto:205-334 (solver-visible fields)
  enum Type {
    UNKNOWN = 0, INDICATOR = 1, CONSTANT = 2, VARIABLE = 3, ADD = 10, RESHAPE = 25,
    LINEAR_MAP = 300, PROX_FUNCTION = 301
  };
  int expression_type = 0;       // 1
  Size size;                     // 2
  std::vector<Expression> arg;   // 3
  Constant constant;             // 8
  std::string variable_id;       // 9{1}
  int cone_type = 0;             // 13{1}  (Cone::ZERO = 1)
  LinearMap linear_map;          // 18
  ProxFunction prox_function;    // 19
};

struct Problem {  // expression.proto:339-346
  Expression objective;                // 1
  std::vector<Expression> constraint;  // 2
};

struct SolverParams {  // solver_params.proto:4-71 (proto2 defaults)
  enum Solver { PROX_ADMM = 0, PROX_ADMM_TWO_BLOCK = 1 };
  int32_t max_iterations = 10000;  // 2
  double rho = 1;                  // 11
  double rel_tol = 1e-2;           // 13
  double abs_tol = 1e-4;           // 14
  int32_t epoch_iterations = 10;   // 18
  bool ignore_stopping_criteria = false;  // 24 (declared by the reference, read only here)
  bool verbose = false;            // 27
  int32_t log_iterations = 100;    // 28
  int solver = PROX_ADMM;          // 30
  bool warm_start = false;         // 31
};

struct SolverStatus {  // solver.proto:4-60 (the fields the reference sets, + timing)
  enum State {
    NOT_STARTED = 0, INITIALIZING = 1, RUNNING = 2, OPTIMAL = 3, MAX_ITERATIONS_REACHED = 4,
    ERROR = 5
  };
  int state = NOT_STARTED;   // 1
  int32_t num_iterations = 0;  // 3
  double total_time = 0, init_time = 0;  // 4{1,2}
  double r_norm = 0, s_norm = 0, epsilon_primal = 0, epsilon_dual = 0;  // 5{1..4}
  std::string Serialize() const;
};

// All Parse* throw eps::Error on malformed input.
Problem ParseProblem(const void* data, size_t len);
Expression ParseExpression(const void* data, size_t len);
LinearMap ParseLinearMap(const void* data, size_t len);
Constant ParseConstant(const void* data, size_t len);
SolverParams ParseSolverParams(const void* data, size_t len);

}  // namespace pb
}  // namespace eps
