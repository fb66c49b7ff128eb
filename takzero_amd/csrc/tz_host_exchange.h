// tz_host_exchange.h — the seam between the self-play driver (tz_host.cpp) and a transport (tz_comm.cpp): the driver packs
// what one move finished and asks for everybody's; it never sees RCCL.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

struct tz_selfplay;

struct HostExchange {
    int rank = 0, world = 1;
    int writer = 0;   // the rank that keeps (and in tz_selfplay_run appends) everybody's lines; < 0: every rank keeps them
    // variable-size all-gather: mine -> all[rank]
    std::function<int(const std::vector<unsigned char>& mine, std::vector<std::vector<unsigned char>>& all)> all_gather;
};

extern "C" int tz_selfplay_set_exchange(tz_selfplay* sp, const HostExchange* x);
