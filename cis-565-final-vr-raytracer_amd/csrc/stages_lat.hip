// stages.hip's traced kernels for small launches: latency-mode traversal (see the note at the top of stages.hip)
#define RT_LAT 1
#include "stages.hip"
