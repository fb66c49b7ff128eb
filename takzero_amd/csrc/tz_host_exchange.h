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

// Observer of the drivers' decisions (tests only: tests/host_over_oracle.cpp checks every move against oracle/host.hpp).
// Not part of the C ABI.
struct tz_state;
struct HostTrace {
    // selfplay, after the move choice and before take_a_step: draws[g] = the uniform draw in [0, 1) that the sampling of game g
    // consumed (NaN where none was), halving[g] = the sequential-halving result before the early-ply override (kind 1)
    std::function<void(const std::vector<double>& draws, const std::vector<uint16_t>& halving, const std::vector<uint16_t>& actions)> chosen;
    // selfplay, after restart_terminal_envs and the completion of the finished games' targets: rows of `amax`
    std::function<void(const std::vector<int8_t>& terminal, const std::vector<tz_state>& states, const std::vector<uint16_t>& moves,
                       const std::vector<float>& policy, const std::vector<int32_t>& count, const std::vector<float>& value,
                       const std::vector<float>& ube, int amax)> completed;
    // reanalyze, after the search: the selected actions and the targets of the batch
    std::function<void(const std::vector<uint16_t>& selected, const std::vector<uint16_t>& moves, const std::vector<float>& policy,
                       const std::vector<int32_t>& count, const std::vector<float>& value, const std::vector<float>& ube, int amax)> reanalyzed;
};
