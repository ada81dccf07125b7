// Private glue between the translation units of libfrx.so (not part of the C ABI).
#pragma once
#include <string>

namespace frx {
// records the text frx_last_error() returns on this thread and hands `code` back (frx_api.cpp)
int set_error(int code, const std::string &msg);
// Host CPU budget of a resident plan (frx_api.cpp): CPUs the process may use (affinity mask and cgroup quota), and this plan's share of them
// when `LOCAL_WORLD_SIZE` ranks of a node and the other shards of a frx_multi job (concurrent_plans_hint: +n before, -n after) spin next to it.
double host_cpu_budget();
int host_cpu_share();
void concurrent_plans_hint(int delta);
}
