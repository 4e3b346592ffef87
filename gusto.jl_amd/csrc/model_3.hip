// model_3.hip -- instantiates the SCP kernels for gusto_model_id 3
#include "launch.hpp"

int gusto_launch_init_m3(gusto_handle h, bool straight) { return launch_init<3>(h, straight); }
int gusto_launch_scp_m3(gusto_handle h, int mode, int max_iter, int force) { return launch_scp<3>(h, mode, max_iter, force); }
