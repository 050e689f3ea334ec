// Library-level entry points (version, last error) of libmentflow_hip.so.
#include "common.h"
#include "common.cpp.inc"
