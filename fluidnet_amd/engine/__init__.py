"""Native engine: ctypes binding of libfluidnet_hip.so and the plan builder that wires UNet / loss / optimiser
steps into flat op lists executed by fnet_plan_run."""
