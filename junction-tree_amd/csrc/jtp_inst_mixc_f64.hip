// One family of message-passing kernels for one storage type, compiled on its own (jtp_kernels.hip.h, "Explicit instantiation lists").
#define JT_INST_TU
#include "jtp_kernels.hip.h"
JT_INST_MIXC(, double)
