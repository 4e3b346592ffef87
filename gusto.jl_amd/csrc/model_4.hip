// model_4.hip -- instantiates the TrajOpt kernels of the internal model variant 4 (common.hpp: GUSTO_TO_*)
#include "launch.hpp"

int gusto_launch_init_m4(gusto_handle h, bool straight) { return launch_init<4>(h, straight); }
int gusto_launch_trajopt_m4(gusto_handle h, int mode, int max_iter) { return launch_trajopt<4>(h, mode, max_iter); }
