// switches.hpp -- every environment switch of liboemgpu in ONE table, parsed ONCE.
//
// None of them is needed in production: they select between engines that the library otherwise chooses by size (so that the tests can
// hold two engines against each other on the same problem), set the knobs of the host-resident upload, or force a fault (fake timeout).
// The table is read at the first use and never again -- no getenv() on the path of a call (35 switches were read by getenv
// at call time inside run_paths) -- except when a test asks for it: oemgpu_reload_switches() (exported, called by tests/conftest.py
// after every monkeypatch.setenv / delenv) parses the environment again.  DESIGN.md section 7b lists every name with its purpose, and
// tests/test_host_api.py::test_every_switch_is_documented holds the two lists against each other.
#pragma once

#include <cstdlib>
#include <cstring>

namespace oemgpu {

// X(name): the switch's environment variable; sw().name.set / .num (atoll of the value; 0 when unset or not a number) / .str
#define OEM_SWITCH_TABLE(X)                                                                                                          \
    /* engine selection: Gram form */                                                                                                \
    X(OEM_NO_COOP) X(OEM_NO_SYMCOOP) X(OEM_NO_ROWCOOP) X(OEM_NO_FUSED) X(OEM_NO_SYM) \
    X(OEM_SYMCOOP_NO_GENERAL) X(OEM_NO_ZERO_COPY)          \
    /* engine selection: p >= n */                                                                                                   \
    X(OEM_WIDE) X(OEM_NO_WIDE) X(OEM_NO_WCOOP) X(OEM_WRES) X(OEM_NO_WRES) X(OEM_WSTREAM) X(OEM_NO_WSTREAM)        \
    X(OEM_WIDE_NO_GROUP_FUSED) X(OEM_WCOOP_ONE_SET) X(OEM_WCOOP_NO_GENERAL) X(OEM_WCOOP_NO_ALIGN) X(OEM_NO_PENALTY_SPLIT) \
    /* moment kernels, sparse x */                                                                                                   \
    X(OEM_SPARSE_GRAM) X(OEM_SPARSE_TILE_ROWS)             \
    /* faults and checks */                                                                                                          \
    X(OEM_WCOOP_FAKE_TIMEOUT) X(OEM_POISON_OUT) X(OEMGPU_LANCZOS_CAP) X(OEM_NO_ONE_XCD) X(OEM_FAKE_XCD_MISMATCH)                                                                \
    /* the host-resident path */                                                                                                     \
    X(OEMGPU_NO_PEER) X(OEMGPU_NO_PENALTY_SPLIT) X(OEMGPU_CACHE_KEEP_BYTES) X(OEMGPU_UPLOAD_THREADS) X(OEMGPU_SLOT_BYTES)            \
    X(OEMGPU_BLOCK_BYTES) X(OEMGPU_RESIDENT_BYTES)

struct Switch {
    bool set = false;
    long long num = 0;
    char str[24] = "";
};
struct Switches {
#define OEM_SW_FIELD(name) Switch name;
    OEM_SWITCH_TABLE(OEM_SW_FIELD)
#undef OEM_SW_FIELD
    unsigned generation = 0;         // bumped by every reload: per-context state derived from the switches (plan caches, the
                                     // persistent engines' back-off) starts over when it changes
};

const Switches &sw();                // api.hip: the table, parsed at first use
void sw_reload();                    // parse the environment again (oemgpu_reload_switches)

}  // namespace oemgpu
