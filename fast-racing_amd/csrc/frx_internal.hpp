// Private glue between the translation units of libfrx.so (not part of the C ABI).
#pragma once
#include <string>

namespace frx {
// records the text frx_last_error() returns on this thread and hands `code` back (frx_api.cpp)
int set_error(int code, const std::string &msg);
}
