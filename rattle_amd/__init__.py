"""rattle_amd: MI355X-native hot path of RATTLE (`cluster` + `correct`)."""
