// model_1.hip -- instantiates the SCP kernels for gusto_model_id 1
#include "launch.hpp"

int gusto_launch_init_m1(gusto_handle h, bool straight) { return launch_init<1>(h, straight); }
int gusto_launch_scp_m1(gusto_handle h, int mode, int max_iter, int force) {
#ifdef GUSTO_WITH_LANE
    if (lane_decomposition(h)) return launch_lane<1>(h, mode, max_iter, force);   // a lane per problem (lane.hpp)
#endif
    return launch_scp<1>(h, mode, max_iter, force);
}
