// errors.hpp -- thread-local last-error slot behind wepp_last_error().
// The reference reports failures by printing and returning 1 / exit(1)
// (src/usher_common.cpp:14-71, src/mutation_annotated_tree.cpp:474,514,533);
// across a C ABI that becomes "return a code, keep the message".
#pragma once
#include <string>

namespace wepp {
int set_error(int code, const std::string& msg);   // returns `code`
}
