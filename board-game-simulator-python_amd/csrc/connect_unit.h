// connect_unit.h -- what the Connect kernels and their launchers are made of besides connect_kernels.hip itself: the
// geometry record the kernels take by value and the launch tuning.  The Makefile hashes THIS header (with bgs_common.h
// and the kernel source) into the Connect unit's id (bgs_kernel_unit_id(0)): counters under profiles/ are quoted as long
// as that id stands, and an edit to the Bounce unit does not move it.
#pragma once

#include <stdint.h>

struct ConnectGeom {
    int h, w, k, nw;   // nw = 64-bit words per plane
};

// Launch tuning of the fused rollouts.  These live here (a header the unit's id hashes) and not in bgs_capi.hip because
// they change what a launch executes: counters taken under one setting must not be quoted for another.
constexpr int kRolloutOpeningBlocks = 3;    // K2o: 4-ply blocks played in lock step before a board joins the refill loop
constexpr int kGamesPerLaneOneWord = 8;     // one-word Connect boards: games per lane a launch aims for (512 per wave at 2^20)
constexpr int kGamesPerLane = 4;            // every other rollout
