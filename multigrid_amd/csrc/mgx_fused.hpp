// Fused black+red sweep (one read and one write of the field per full sweep).
#pragma once

#include "mgx_common.hpp"

template <typename T>
static bool mgx_fused_supported(const MgxGeom &) { return false; }

template <typename T>
static int mgx_fused_launch(const T *, T *, const MgxGeom &, const MgxSweepParams<T> &, const MgxBox *, int, int,
                            u64 *, hipStream_t) { return -1; }
