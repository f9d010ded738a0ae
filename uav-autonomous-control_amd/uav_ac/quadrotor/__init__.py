"""uav_ac.quadrotor -- drop-in module path of the reference package; the code computes on the GPU through libuavac.so."""
