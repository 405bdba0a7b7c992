// stages.hip with the procedural sun & sky environment compiled in (see the note at the top of stages.hip)
#define RT_SKY 1
#include "stages.hip"
