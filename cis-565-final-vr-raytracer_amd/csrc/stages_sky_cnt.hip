// stages.hip with sun & sky and the counters compiled in (see the note at the top of stages.hip)
#define RT_SKY 1
#define RT_COUNT 1
#include "stages.hip"
