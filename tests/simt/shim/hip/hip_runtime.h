// Stand-in for <hip/hip_runtime.h> when the kernel sources are compiled for the SIMT emulator (tests/simt/simt_emu.h).  Test infrastructure.
#pragma once
#include <type_traits>
#include "simt_emu.h"
