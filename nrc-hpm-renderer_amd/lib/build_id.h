#define NRC_BUILD_ID "c6f6dc6ad0098bc7"
