// stages.hip with the counters of rt_set_counting flushed (see the note at the top of stages.hip)
#define RT_COUNT 1
#include "stages.hip"
