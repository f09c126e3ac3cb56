// One group of the heavy kernel instantiations (see x265_amd_inst.h); compiled with -DXA_PART=1 .. 7.
#include "me_kernels.h"
#include "me_lowres.h"
#include "me_field.h"
#include "me_lane.h"

#ifndef XA_PART
#error "compile with -DXA_PART=1..7"
#endif
#define XA_INST template
#include "x265_amd_inst.h"
