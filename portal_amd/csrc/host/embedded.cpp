// embedded.cpp -- the fixed device sources, embedded at build time (see embed_sources.py).
#include "codegen.h"
#include "embedded_device_sources.inc"
