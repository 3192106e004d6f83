// IR sub-tree -> (BlockMatrix A, BlockVector b) under a row key.
// Same contract as the reference (src/epsilon/affine/affine.h:21-45, affine.cc:22-140).
#pragma once

#include <map>
#include <string>

#include "block.h"
#include "wire.h"

namespace eps {
namespace affine {

std::string constraint_key(int i);  // "constraint:<i>"
std::string arg_key(int i);         // "arg:<i>"

// A may be null (only the constant part is wanted), b may be null.
void BuildAffineOperator(const pb::Expression& expr, DataMap* data, const std::string& row_key,
                         BlockMatrix* A, BlockVector* b);

}  // namespace affine

int64_t GetDimension(const pb::Expression& expr);  // expression_util.cc:50-53
// Variables of an expression ordered by id (expression_util.cc:11-31).
void GetVariables(const pb::Expression& expr, std::map<std::string, const pb::Expression*>* vars);
std::map<std::string, const pb::Expression*> GetVariables(const pb::Problem& problem);

}  // namespace eps
