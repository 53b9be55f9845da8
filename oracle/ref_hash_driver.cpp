// Calls the REFERENCE's own photonCell / photonBucket (photon_hash.cuh:10-24) over arrays, so that their
// outputs can be written down as fixtures.  Test infrastructure: built only in the container that holds
// /root/reference (oracle/Makefile, target _ref), into oracle/_ref/, with the reference's header compiled
// unmodified from where it lies and NVIDIA's own cuda_runtime.h from the include directory the Makefile
// names (the one that ships with the triton package of this image).  Nothing of the product links or
// loads it; the fixtures made with it are what travels (tests/golden/make_hash_fixture.py).
//
// The header's functions are __device__ functions; under a host compiler NVIDIA's host_defines.h turns
// that into an attribute g++ ignores, and floorf / the float division are the host's IEEE ones
// (-ffp-contract=off, no fast-math).  static_cast<int>(float) out of int's range is undefined on the
// host, so callers pass only positions with |p / cellSize| < 2^31.
#include <math.h>  // floorf in the global namespace, which nvcc would have declared before the header
#include <cstddef>
#include <cstdint>

#include <photon_hash.cuh>

extern "C" {

// cells: n x 3 int32; keys: n uint32.
void ref_photon_hash(const float* positions, size_t n, float cellSize, unsigned int bucketCount, int32_t* cells,
                     uint32_t* keys) {
    for (size_t i = 0; i < n; ++i) {
        const float3 p = make_float3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
        const int3 cell = rt::photonCell(p, cellSize);
        cells[3 * i] = cell.x;
        cells[3 * i + 1] = cell.y;
        cells[3 * i + 2] = cell.z;
        keys[i] = rt::photonBucket(cell, bucketCount);
    }
}

}
