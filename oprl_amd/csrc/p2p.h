// p2p.h — state of the one-shot peer-window all-reduce (csrc/p2p.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace oprl {

constexpr int kP2pMaxWorld = 8;
constexpr int kP2pBlocks = 64, kP2pThreads = 256;
constexpr size_t kFlagStride = 16;           // uint64 per flag slot: one 128-byte line each

struct P2pState {
  int world = 0, rank = 0;
  bool connected = false;
  size_t slot_floats = 0;                    // capacity of one (parity, source) slot
  size_t window_bytes = 0;
  char* window = nullptr;                    // this rank's window
  char* peer[kP2pMaxWorld] = {nullptr};      // every rank's window as mapped here (peer[rank] == window)
  unsigned* done = nullptr;                  // device counter of finished push workgroups
  unsigned long long seq = 0;                // exchanges so far
  size_t tile_off = 0, tile_bytes = 0;       // second region of the window: k_dw_adam<true>'s per-tile exchange
  unsigned long long tile_seq = 0;
  unsigned* err = nullptr;                   // the learner's host-visible error word (an expired flag wait is reported there)
};

hipError_t p2p_create(P2pState& s, int rank, int world, size_t max_floats, size_t tile_region_bytes, void* handle_out);
hipError_t p2p_connect(P2pState& s, const void* handles);
void p2p_destroy(P2pState& s);
hipError_t p2p_all_reduce(P2pState& s, void* buf, size_t n, bool as_double, hipStream_t st);

}  // namespace oprl
