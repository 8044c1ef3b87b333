// ABI version of libmqdet_hip.so (see include/mqdet_hip.h).
#include "../../include/mqdet_hip.h"
extern "C" int mq_abi_version(void) { return 31; }
