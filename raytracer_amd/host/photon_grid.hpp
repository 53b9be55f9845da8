// The reference's src/photon_grid.hpp:12-20 over the engine's C ABI: same two
// functions and argument lists; the status is an int (0 = success, otherwise the
// code of the first failing call) where the reference returns cudaError_t.
#pragma once

#include <cstddef>

#include "rt_engine.h"

namespace rt {

using Photon = rt_photon;

inline std::size_t photonScanStorage(unsigned int bucketCount) {
    return rt_photon_scan_storage(bucketCount);
}

inline int buildPhotonGrid(const Photon* photons, unsigned int count, float cellSize, unsigned int bucketCount,
                           unsigned int* counts, unsigned int* offsets, void* scan, std::size_t scanBytes,
                           Photon* sorted) {
    return rt_build_photon_grid(photons, count, cellSize, bucketCount, counts, offsets, scan, scanBytes, sorted);
}

}  // namespace rt
