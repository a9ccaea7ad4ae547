#include "errors.hpp"

#include "../../include/wepp_place.h"

namespace wepp {
static thread_local std::string g_last_error;
int set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace wepp

extern "C" const char* wepp_last_error(void) { return wepp::g_last_error.c_str(); }
