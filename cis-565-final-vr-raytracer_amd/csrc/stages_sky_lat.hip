// stages.hip's traced kernels for small launches with sun & sky compiled in (see the note at the top of stages.hip)
#define RT_SKY 1
#define RT_LAT 1
#include "stages.hip"
