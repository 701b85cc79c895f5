"""Step engines: the device-side plan of one ``train_one_step`` per algorithm."""
