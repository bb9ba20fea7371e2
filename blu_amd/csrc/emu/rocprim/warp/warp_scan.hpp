// emu stand-in for <rocprim/warp/warp_scan.hpp>: blu_dev.h reduces through the emulator's collective instead.
#pragma once
